// mfma_i8_layout.hip -- what k_blur's matrix-core form relies on, checked on the device (round 6):
//   v_mfma_i32_16x16x32_i8:  D[m][n] = C[m][n] + sum over lane groups g = 0..3 and bytes j = 0..7 of
//                            A(lane 16 g + m).byte j  x  B(lane 16 g + n).byte j          (signed bytes)
//   with D / C in lane 16 g' + n, register i  <->  m = 4 g' + i                            (cdna_hip_programming.md: C/D layout)
//   v_mfma_i32_16x16x64_i8: the same with bytes j = 0..15; and the 4 x 4 transpose over (register, lane group) by v_permlane swaps.
// i.e. the K index is (lane group, byte) on BOTH operands alike -- which K that is never matters to a caller that builds A and B
// with the same rule.  Also times a chain of dependent-free MFMAs per wave.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_i8_layout tools/c/mfma_i8_layout.hip && ./mfma_i8_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k_one(const long* a, const long* b, const v4i* c, v4i* d)
{
    d[threadIdx.x] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a[threadIdx.x], b[threadIdx.x], c[threadIdx.x], 0, 0, 0);
}

// the K = 64 form pass 1 uses (16 bytes a lane), and the 4 x 4 transpose over (register, lane group) by row swaps
__global__ void k_one64(const v4i* a, const v4i* b, const v4i* c, v4i* d)
{
    d[threadIdx.x] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], c[threadIdx.x], 0, 0, 0);
}
__global__ void k_transpose(const unsigned* in, unsigned* out)
{
    unsigned r[4];
    for (int k = 0; k < 4; k++) r[k] = in[64 * k + threadIdx.x];
    auto s0 = __builtin_amdgcn_permlane32_swap(r[0], r[2], false, false); r[0] = s0[0]; r[2] = s0[1];
    auto s1 = __builtin_amdgcn_permlane32_swap(r[1], r[3], false, false); r[1] = s1[0]; r[3] = s1[1];
    auto s2 = __builtin_amdgcn_permlane16_swap(r[0], r[1], false, false); r[0] = s2[0]; r[1] = s2[1];
    auto s3 = __builtin_amdgcn_permlane16_swap(r[2], r[3], false, false); r[2] = s3[0]; r[3] = s3[1];
    for (int k = 0; k < 4; k++) out[64 * k + threadIdx.x] = r[k];
}

__global__ void k_rate(const long* a, const long* b, v4i* d, int iters)
{
    const long av = a[threadIdx.x & 63], bv = b[threadIdx.x & 63];
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(av, bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(av, bv, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(av, bv, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(av, bv, c3, 0, 0, 0);
    }
    d[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
}

int main()
{
    signed char A[64][8], B[64][8];
    int C[64][4], D[64][4], R[64][4];
    srand(5);
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 8; j++) { A[l][j] = (signed char)(rand() % 256 - 128); B[l][j] = (signed char)(rand() % 256 - 128); }
    for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) C[l][i] = rand() % 100000 - 50000;
    for (int gp = 0; gp < 4; gp++)
        for (int n = 0; n < 16; n++)
            for (int i = 0; i < 4; i++) {
                const int m = 4 * gp + i;
                long s = C[16 * gp + n][i];
                for (int g = 0; g < 4; g++)
                    for (int j = 0; j < 8; j++) s += (long)A[16 * g + m][j] * B[16 * g + n][j];
                R[16 * gp + n][i] = (int)s;
            }
    long *da, *db; v4i *dc, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 1024); hipMalloc(&dd, 1024 * 1024 * 16);
    hipMemcpy(da, A, 512, hipMemcpyHostToDevice); hipMemcpy(db, B, 512, hipMemcpyHostToDevice); hipMemcpy(dc, C, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
    hipMemcpy(D, dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) bad += D[l][i] != R[l][i];
    printf("layout: %d of 256 accumulator entries differ from the (lane group, byte) rule -> %s\n", bad, bad ? "WRONG" : "ok");
    {   // K = 64: the same rule with 16 bytes a lane
        signed char A6[64][16], B6[64][16];
        int R6[64][4], D6[64][4];
        for (int l = 0; l < 64; l++)
            for (int j = 0; j < 16; j++) { A6[l][j] = (signed char)(rand() % 256 - 128); B6[l][j] = (signed char)(rand() % 256 - 128); }
        for (int gp = 0; gp < 4; gp++)
            for (int n = 0; n < 16; n++)
                for (int i = 0; i < 4; i++) {
                    const int m = 4 * gp + i;
                    long s = C[16 * gp + n][i];
                    for (int g = 0; g < 4; g++)
                        for (int j = 0; j < 16; j++) s += (long)A6[16 * g + m][j] * B6[16 * g + n][j];
                    R6[16 * gp + n][i] = (int)s;
                }
        v4i *da6, *db6;
        hipMalloc(&da6, 1024); hipMalloc(&db6, 1024);
        hipMemcpy(da6, A6, 1024, hipMemcpyHostToDevice); hipMemcpy(db6, B6, 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_one64, dim3(1), dim3(64), 0, 0, da6, db6, dc, dd);
        hipMemcpy(D6, dd, 1024, hipMemcpyDeviceToHost);
        int bad6 = 0;
        for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) bad6 += D6[l][i] != R6[l][i];
        printf("layout, 16x16x64: %d of 256 accumulator entries differ from the (lane group, byte) rule -> %s\n", bad6, bad6 ? "WRONG" : "ok");
        bad += bad6;
        // transpose: register k of lane group g  <->  register g of lane group k
        unsigned T[4][64], U[4][64];
        for (int k = 0; k < 4; k++) for (int l = 0; l < 64; l++) T[k][l] = (unsigned)(1000 * k + l);
        unsigned *dt, *du;
        hipMalloc(&dt, sizeof T); hipMalloc(&du, sizeof U);
        hipMemcpy(dt, T, sizeof T, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_transpose, dim3(1), dim3(64), 0, 0, dt, du);
        hipMemcpy(U, du, sizeof U, hipMemcpyDeviceToHost);
        int badt = 0;
        for (int k = 0; k < 4; k++) for (int l = 0; l < 64; l++) badt += U[k][l] != T[l >> 4][16 * k + (l & 15)];
        printf("transpose by v_permlane32_swap / v_permlane16_swap: %d of 256 entries wrong -> %s\n", badt, badt ? "WRONG" : "ok");
        bad += badt;
    }
    // rate: 1024 blocks x 256 threads (4 waves per block = one per SIMD of a CU, 4 blocks per CU), 4 x iters MFMAs per wave
    const int iters = 20000, blocks = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, da, db, dd, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, da, db, dd, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 * 4 * iters / 1024.0;     // MFMAs per SIMD (1024 SIMDs)
    printf("rate: %.3f ms for %.0f MFMAs per SIMD -> %.1f ns each = %.1f cycles at 2.4 GHz\n", ms, per_simd, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    return bad != 0;
}
