// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the convolution kernels
// (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half of the bytes of wide coalesced 16 B/lane reads; "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel reads (or writes) each byte of a 2 GiB array exactly once, 16 bytes per lane; what differs is how the lanes
// of a wave are laid out: SEG contiguous bytes per segment, consecutive segments of a wave STRIDE bytes apart (an NCHW row
// piece of a tile: 64 B = 16 fp32 pixels, 128 / 256 B = 32 / 64 pixels, 1024 = fully coalesced).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- /tmp/fetch_calib   (then WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// the array is a [rows][ROWB bytes] matrix; a wave reads 64 lanes x 16 B = 1024 B as (1024 / SEG) segments taken from
// consecutive rows at the same column block
template <int SEG, bool WRITE>
__global__ __launch_bounds__(256) void seg_kernel(float* __restrict__ p, long rows, int rowb, float* __restrict__ sink) {
    constexpr int LPS = SEG / 16;                  // lanes per segment
    constexpr int SPW = 64 / LPS;                  // segments (= rows) per wave-load
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const int cols = rowb / SEG;                   // column blocks per row
    const long units = (rows / SPW) * cols;        // wave-loads in the array
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long u = wave; u < units; u += nwaves) {
        const long rb = u / cols, cb = u - rb * cols;
        const long row = rb * SPW + lane / LPS;
        char* q = reinterpret_cast<char*>(p) + row * (long)rowb + cb * SEG + (lane % LPS) * 16;
        if (WRITE) *reinterpret_cast<f32x4*>(q) = f32x4{1.f, 2.f, 3.f, (float)lane};
        else acc += *reinterpret_cast<const f32x4*>(q);
    }
    if (!WRITE && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <int SEG, bool WRITE> __global__ void dummy() {}
template <int SEG>
void run(float* p, long bytes, int rowb, float* sink) {
    const long rows = bytes / rowb;
    hipLaunchKernelGGL((seg_kernel<SEG, false>), dim3(256 * 16), dim3(256), 0, 0, p, rows, rowb, sink);
    hipLaunchKernelGGL((seg_kernel<SEG, true>), dim3(256 * 16), dim3(256), 0, 0, p, rows, rowb, sink);
    hipDeviceSynchronize();
}

int main() {
    const long bytes = 2L << 30;
    float *p, *sink;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&sink, 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(p, 0, bytes);
    hipDeviceSynchronize();
    const int rowb = 1024;                         // a 256-pixel fp32 row (512-pixel bf16 row)
    for (int rep = 0; rep < 2; ++rep) {
        run<64>(p, bytes, rowb, sink);
        run<128>(p, bytes, rowb, sink);
        run<256>(p, bytes, rowb, sink);
        run<1024>(p, bytes, rowb, sink);
    }
    printf("each seg_kernel<SEG, false> launch reads and each seg_kernel<SEG, true> launch writes %ld bytes\n", bytes);
    return 0;
}
