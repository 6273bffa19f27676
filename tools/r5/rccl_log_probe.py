import os, torch, torch.distributed as dist, re
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1', NCCL_DEBUG='INFO', NCCL_DEBUG_SUBSYS='INIT', NCCL_DEBUG_FILE='/tmp/rccl0.log', HSA_ENABLE_IPC_MODE_LEGACY='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
x = torch.ones(1024, device='cuda'); dist.all_reduce(x); torch.cuda.synchronize()
txt = open('/tmp/rccl0.log').read()
print(len(txt)); print([l for l in txt.splitlines() if 'hannel' in l][:12])
print(re.findall(r'(\d+) coll channels', txt), re.findall(r'Channel (\d+)/(\d+)', txt)[:3])
dist.destroy_process_group()
