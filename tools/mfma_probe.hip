// Layout and rate probe of v_mfma_f64_16x16x4_f64 on gfx950 (the reduced system of the column pass, hadi_pb_reduced_mfma):
//   hipcc -O3 --offload-arch=gfx950 -o tools/mfma_probe tools/mfma_probe.hip && tools/mfma_probe
// Prints whether the operand / result mapping the kernels assume holds, and the instruction's issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe(const double *A /*16x4*/, const double *B /*4x16*/, double *D /*16x16*/) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];   // assumed: lane 16 k + i holds A[i][k]
    const double b = B[(l >> 4) * 16 + (l & 15)];  // assumed: lane 16 k + j holds B[k][j]
    d4 c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * r + (l >> 4)) * 16 + (l & 15)] = c[r];  // lane 16 q + j, register r holds D[4 r + q][j] (found below)
}
// raw: lane l gets a = Araw[l], b = Braw[l]; every lane's four result registers come back as Draw[l][r]
__global__ void raw(const double *Araw, const double *Braw, double *Draw) {
    const int l = threadIdx.x;
    d4 c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(Araw[l], Braw[l], c, 0, 0, 0);
    for (int r = 0; r < 4; r++) Draw[l * 4 + r] = c[r];
}
__global__ void rate(double *out, int iters) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void rate_dep(double *out, int iters) {  // one dependent chain per wavefront
    d4 c0 = {0, 0, 0, 0};
    const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < 4 * iters; i++) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0];
}
int main() {
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1.0 + i + 0.01 * k;
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 0.5 + 0.1 * k - 0.03 * j;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dD, 1 << 24);
    hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
    double err = 0; for (int e = 0; e < 256; e++) err = fmax(err, fabs(D[e] - R[e]));
    printf("layout: max |D - A B| = %.3e  (%s)\n", err, err < 1e-12 ? "the assumed mapping holds" : "MAPPING WRONG");
    {   // find the mapping: a(l) = 2^(l) pattern is too wide for doubles; use two runs with one-hot operands instead
        // run 1: a(l) = 1 + l, b(l) = 1 for lanes with the same "k" -- brute force over the candidate index maps on the host
        std::vector<double> Ar(64), Br(64), Dr(256);
        for (int l = 0; l < 64; l++) { Ar[l] = 1.0 + l; Br[l] = 100.0 + 7.0 * l; }
        hipMemcpy(dA, Ar.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, Br.data(), 64 * 8, hipMemcpyHostToDevice);
        raw<<<1, 64>>>(dA, dB, dD);
        hipMemcpy(Dr.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
        // candidates: A lane->(i,k): (l%16, l/16) or (l/4, l%4); B lane->(k,j): (l/16, l%16) or (l%4, l/4);
        // D (lane, r)->(i,j): {i = 4 (l/16) + r, j = l%16}, {i = l%16, j = 4 (l/16) + r}, {i = l/4 ... }
        for (int ca = 0; ca < 2; ca++) for (int cb = 0; cb < 2; cb++) for (int cd = 0; cd < 4; cd++) {
            double Am[16][4], Bm[4][16];
            for (int l = 0; l < 64; l++) {
                const int ia = ca ? l / 4 : l % 16, ka = ca ? l % 4 : l / 16;
                const int kb = cb ? l % 4 : l / 16, jb = cb ? l / 4 : l % 16;
                Am[ia][ka] = Ar[l]; Bm[kb][jb] = Br[l];
            }
            double e = 0;
            for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
                int i, j;
                if (cd == 0) { i = 4 * (l / 16) + r; j = l % 16; }
                else if (cd == 1) { i = l % 16; j = 4 * (l / 16) + r; }
                else if (cd == 2) { i = 4 * r + l / 16; j = l % 16; }
                else { i = l % 16; j = 4 * r + l / 16; }
                double want = 0; for (int k = 0; k < 4; k++) want += Am[i][k] * Bm[k][j];
                e = fmax(e, fabs(Dr[l * 4 + r] - want));
            }
            printf("A map %d, B map %d, D map %d: max err %.3e%s\n", ca, cb, cd, e, e < 1e-9 ? "   <== holds" : "");
        }
    }
    for (int which = 0; which < 2; which++) for (int wpb : {256, 512, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000, blocks = 256;
        if (which) rate_dep<<<blocks, wpb>>>(dD, 10); else rate<<<blocks, wpb>>>(dD, 10);
        hipEventRecord(e0);
        if (which) rate_dep<<<blocks, wpb>>>(dD, iters); else rate<<<blocks, wpb>>>(dD, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)blocks * (wpb / 64) * 4.0 * iters;  // MFMA instructions
        printf("%s, %4d threads/block x 256 blocks: %.2f TFLOP/s fp64, %.1f ns per MFMA per CU-wave-slot (%.3f ms)\n", which ? "one dependent chain " : "4 independent chains",
               wpb, n * 2048.0 / (ms * 1e-3) / 1e12, ms * 1e6 / (4.0 * iters), ms);
    }
    return 0;
}
