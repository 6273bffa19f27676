"""Which entry points of include/aide_hip.h does the product reach?  Run the GPU tests and bench.py with
AIDE_ABI_COVERAGE=gpurun_out/cov (one JSON per process), then: python tools/abi_coverage.py gpurun_out/cov > profiles/rNN_abi_coverage.md"""
import glob
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aide_amd._lib import parse_header


def main():
    prefix = sys.argv[1]
    counts = {}
    files = glob.glob(prefix + '.*')
    for f in files:
        for k, v in json.load(open(f)).items():
            counts[k] = counts.get(k, 0) + v
    protos = parse_header()
    never = sorted(n for n in protos if n not in counts)
    print('# C-ABI coverage: %d of %d entry points of include/aide_hip.h reached (%d processes)\n' % (
        len(protos) - len(never), len(protos), len(files)))
    print('Not reached by `pytest -m gpu` + `bench.py` (all workloads):\n')
    for n in never:
        print('* `%s`' % n)
    print('\nReached (Python-side calls; tape replays re-issue the recorded launches without passing here):\n')
    for n in sorted(counts, key=lambda k: -counts[k]):
        print('* `%s` %d' % (n, counts[n]))


if __name__ == '__main__':
    main()
