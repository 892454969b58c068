// Measures the issue rate of integer VALU instructions on one SIMD as the number of resident waves grows.
// Prints, per (op, waves per SIMD), the cycles one SIMD spends per wave64 instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(256) void probe(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = seed * (k + 1) + threadIdx.x;
    uint32_t b = seed ^ threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (OP == 0) a[k] = min(a[k] ^ b, a[(k + 1) & 7]);             // v_xor + v_min_u32
                if (OP == 1) a[k] = __popc(a[k] ^ b) + a[(k + 1) & 7];         // v_xor + v_bcnt (with add)
                if (OP == 2) a[k] = __builtin_amdgcn_udot4(a[k], b, a[(k + 1) & 7], false);  // v_dot4_u32_u8
                if (OP == 3) a[k] = min(min(a[k], b), a[(k + 1) & 7]) + 1;     // v_min3_u32 + v_add
                if (OP == 5) a[k] = a[k] * b + a[(k + 1) & 7];                 // v_mul_lo_u32 + v_add (or v_mad_u64_u32)
                if (OP == 6) a[k] = __mul24((int)a[k], (int)b) + a[(k + 1) & 7];  // v_mad_i32_i24
                if (OP == 4) {                                                 // v_fma_f32
                    float x = __uint_as_float(a[k]), y = __uint_as_float(a[(k + 1) & 7]);
                    a[k] = __float_as_uint(__builtin_fmaf(x, 1.0001f, y));
                }
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s ^= a[k];
    if (s == 0x12345678u) out[0] = s;
}

typedef float float2v __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void probe2(uint32_t* out, int iters, float seed) {
    float2v a[8];
    double da[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        a[k] = float2v{seed * (k + 1) + threadIdx.x, seed + k};
        da[k] = seed * (k + 1) + threadIdx.x;
    }
    const float2v m = {1.0001f, 0.9999f};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (OP == 0) a[k] = a[k] * m;                                  // v_pk_mul_f32
                if (OP == 1) a[k] = a[k] + a[(k + 1) & 7];                     // v_pk_add_f32
                if (OP == 2) da[k] = __builtin_fma(da[k], 1.0000001, da[(k + 1) & 7]);  // v_fma_f64
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += a[k].x + a[k].y + (float)da[k];
    if (s == 0.12345f) out[0] = 1;
}

template <int OP>
static void run(const char* name, int instrPerInner, uint32_t* d) {
    const int iters = 2048;
    for (int wps = 1; wps <= 8; wps *= 2) {
        // one workgroup of 256 threads = one wave per SIMD of a CU; wps workgroups per CU
        dim3 grid(256 * wps), block(256);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<OP>, grid, block, 0, 0, d, iters, 12345u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<OP>, grid, block, 0, 0, d, iters, 12345u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instrPerWave = (double)iters * 64 * instrPerInner;
        const double perSimd = instrPerWave * wps;
        printf("%-22s waves/SIMD %d  %.3f ms  %.2f ns per wave64 instr per SIMD (%.2f cyc @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / perSimd, ms * 1e6 / perSimd * 2.4);
    }
}

template <int OP>
static void run2(const char* name, uint32_t* d) {
    const int iters = 2048;
    for (int wps = 1; wps <= 8; wps *= 2) {
        dim3 grid(256 * wps), block(256);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(probe2<OP>, grid, block, 0, 0, d, iters, 1.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe2<OP>, grid, block, 0, 0, d, iters, 1.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double perSimd = (double)iters * 64 * wps;
        printf("%-22s waves/SIMD %d  %.3f ms  %.2f ns per wave64 instr per SIMD (%.2f cyc @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / perSimd, ms * 1e6 / perSimd * 2.4);
    }
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 4096);
    run<0>("xor+min_u32", 2, d);
    run<1>("xor+bcnt(add)", 2, d);
    run<2>("dot4_u32_u8", 1, d);
    run<3>("min3+add", 2, d);
    run<4>("fma_f32", 1, d);
    run<5>("mul_lo_u32+add", 2, d);
    run<6>("mad_i32_i24", 1, d);
    run2<0>("pk_mul_f32", d);
    run2<1>("pk_add_f32", d);
    run2<2>("fma_f64", d);
    hipFree(d);
    return 0;
}
