// vmem_rate.hip -- what a vector memory load costs a CU's texture path on gfx950, by width and by active lanes: every lane reads
// window-like rows (pitch 768 B, 16-byte segments) of an L2-resident image; 8 waves per SIMD; prints ns and bytes per CU clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 vmem_rate.hip -o vmem_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REPS 256
// PAT: 0 = rows of 3 segments from a 4-byte-aligned start (the window staging of k_orient_desc), 1 = the same from a 16-byte-aligned
// start, 2 = rows of 4 segments from a 64-byte-aligned start, 3 = all 64 lanes contiguous (1 KB), 4 = rows of 2 segments, 16-byte-aligned
template <int W, int ACTIVE, int PAT = 0>   // W = dwords per lane (1, 2, 3, 4), ACTIVE = lanes out of 64 that load
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ img, unsigned* out, int pitch, int span)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    unsigned acc = 0;
    const int per = PAT == 2 ? 4 : (PAT == 3 ? 64 : (PAT == 4 ? 2 : 3));
    const int row = lane / per, seg = lane % per;
    const unsigned al = PAT == 0 ? 4u : (PAT == 2 || PAT == 3 ? 64u : 16u);
    unsigned base = ((unsigned)((wave * 37) % span) * 4u) / al * al + (unsigned)(row * pitch + seg * 16);
    if (lane < ACTIVE) {
        for (int r = 0; r < REPS; r++) {
            const uint8_t* p = img + base;
            if (W == 4) { const uint4 v = *reinterpret_cast<const uint4*>(p); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
            if (W == 3) { const uint3 v = *reinterpret_cast<const uint3*>(p); acc ^= v.x ^ v.y ^ v.z; }
            if (W == 2) { const uint2 v = *reinterpret_cast<const uint2*>(p); acc ^= v.x ^ v.y; }
            if (W == 1) { acc ^= *reinterpret_cast<const unsigned*>(p); }
            base += (unsigned)(pitch * 10 + al * (acc & 1u));   // next block of rows (the dependence keeps the loads from being merged)
            if (base > (unsigned)(span * 4)) base -= (unsigned)(span * 4);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int W, int ACTIVE, int PAT = 0> static void run(const uint8_t* d, unsigned* o, int pitch, int span)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * 8;   // 8 waves per SIMD resident
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k<W, ACTIVE, PAT>), dim3(blocks), dim3(256), 0, 0, d, o, pitch, span);
    (void)hipDeviceSynchronize(); (void)hipEventRecord(a, 0);
    const int L = 10;
    for (int w = 0; w < L; w++) hipLaunchKernelGGL((k<W, ACTIVE, PAT>), dim3(blocks), dim3(256), 0, 0, d, o, pitch, span);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)blocks * 4 * REPS * L / 256.0;
    const double ns = ms * 1e6 / instr_per_cu;
    printf("pattern %d W=%d dwords, %2d lanes: %7.3f ms  %6.2f ns per wave-load and CU = %5.1f clk at 2.4 GHz, %5.1f B/clk\n", PAT, W, ACTIVE, ms, ns, ns * 2.4, W * 4.0 * ACTIVE / (ns * 2.4));
}
int main()
{
    const int pitch = 768, rows = 1400; const size_t n = (size_t)pitch * rows;
    uint8_t* d; unsigned* o; (void)hipMalloc(&d, n + 4096); (void)hipMalloc(&o, 64); (void)hipMemset(d, 1, n + 4096);
    const int span = (int)((n - (size_t)pitch * 40) / 4);
    run<4, 64>(d, o, pitch, span); run<4, 32>(d, o, pitch, span); run<4, 48>(d, o, pitch, span);
    run<4, 64, 1>(d, o, pitch, span); run<4, 64, 2>(d, o, pitch, span); run<4, 64, 3>(d, o, pitch, span); run<4, 64, 4>(d, o, pitch, span);
    run<3, 64>(d, o, pitch, span); run<2, 64>(d, o, pitch, span); run<2, 32>(d, o, pitch, span); run<1, 64>(d, o, pitch, span);
    return 0;
}
