// Pins the unit of the SQ_LDS_BANK_CONFLICT counter (and of SQ_ACTIVE_INST_LDS / SQ_LDS_IDX_ACTIVE beside it).
// Every wave issues a known number of ds_read_b32 whose lane addresses are lane * STRIDE dwords:
//   STRIDE 1: conflict-free (32 lanes of a group on 32 different banks)
//   STRIDE 2: 2-way (lanes i and i+16 of a group share a bank, different dwords)
//   STRIDE 4: 4-way,  STRIDE 8: 8-way,  STRIDE 32: 32-way
// plus MODE 1: ds_read_u8 at pseudo-random byte addresses (the access shape of an exact-strength batch).
// Run one configuration per process under `rocprofv3 --pmc ...`: the counters divided by the number of wave
// instructions (printed) give extra cycles per instruction, in whatever unit the counter has.
//   usage: lds_conflict_probe <stride> <mode> [waves_per_block]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

constexpr int kIters = 2000;      // loop trips per wave
constexpr int kUnroll = 16;       // reads per trip

template <int MODE>
__global__ __launch_bounds__(256) void probe(uint32_t* out, int stride, uint32_t seed) {
    __shared__ uint32_t lds[64 * 32 + 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 32 + 64; i += blockDim.x) lds[i] = i * 2654435761u + seed;
    __syncthreads();
    uint32_t acc = 0;
    if (MODE == 0) {
        const uint32_t* p = lds + lane * stride;
        for (int it = 0; it < kIters; it++) {
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                uint32_t v;
                // offsets are multiples of 128 bytes (32 banks): the same bank pattern in every read
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((uint32_t)(uintptr_t)p), "n"(0));
                acc += v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else {
        // pseudo-random byte address per lane, fixed over the run (rows of 48 bytes, as a FAST tile)
        uint32_t h = (lane + 1) * 2246822519u ^ seed;
        h ^= h >> 15; h *= 2654435761u; h ^= h >> 13;
        const int y = 3 + (int)(h % 32u), x = 4 + (int)((h >> 8) % 32u);
        const uint8_t* p = reinterpret_cast<const uint8_t*>(lds) + y * 48 + x;
        for (int it = 0; it < kIters; it++) {
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                uint32_t v;
                asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)p));
                acc += v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
    const int stride = argc > 1 ? atoi(argv[1]) : 1;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const int waves = argc > 3 ? atoi(argv[3]) : 4;
    if (stride < 1 || stride > 32 || waves < 1 || waves > 4) {
        fprintf(stderr, "stride 1..32, waves per block 1..4\n");
        return 2;
    }
    uint32_t* out = nullptr;
    if (hipMalloc(&out, 256) != hipSuccess) return 1;
    const int blocks = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(64 * waves), 0, 0, out, stride, 1u);
    else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(64 * waves), 0, 0, out, stride, 1u);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * waves * kIters * kUnroll;
    printf("{\"stride\": %d, \"mode\": %d, \"waves_per_block\": %d, \"lds_wave_instructions\": %.0f, \"ms\": %.3f, "
           "\"ns_per_inst_per_cu\": %.3f}\n", stride, mode, waves, insts, ms, ms * 1e6 / (insts / 256));
    hipFree(out);
    return 0;
}
