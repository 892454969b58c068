// Probe: do v_pk_minimum3_f16 / v_pk_maximum3_f16 order the half-float subnormals 0x0000..0x00FF like integers,
// and do their operand-select bits exchange the halves of an operand?  Prints mismatches against the integer model.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/pk_minmax3_probe.hip -o tools/probe/pk_minmax3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = a[i], y = b[i], z = c[i], r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z)); o[6 * i + 0] = r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(x), "v"(y), "v"(z)); o[6 * i + 1] = r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(x), "v"(y), "v"(z)); o[6 * i + 2] = r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z)); o[6 * i + 3] = r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(x), "v"(y), "v"(z)); o[6 * i + 4] = r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(x), "v"(y), "v"(z)); o[6 * i + 5] = r;
}
int main() {
    const int n = 1 << 16;
    std::vector<uint32_t> a(n), b(n), c(n), o(6 * n);
    uint64_t s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
    for (int i = 0; i < n; i++) {
        a[i] = (rnd() & 0xFF) | (rnd() & 0xFF) << 16;
        b[i] = (rnd() & 0xFF) | (rnd() & 0xFF) << 16;
        c[i] = (rnd() & 0xFF) | (rnd() & 0xFF) << 16;
    }
    uint32_t *da, *db, *dc, *dout;
    hipMalloc(&da, 4 * n); hipMalloc(&db, 4 * n); hipMalloc(&dc, 4 * n); hipMalloc(&dout, 24 * n);
    hipMemcpy(da, a.data(), 4 * n, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), 4 * n, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), 4 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
    if (hipMemcpy(o.data(), dout, 24 * n, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    auto lo = [](uint32_t v) { return v & 0xFFFFu; };
    auto hi = [](uint32_t v) { return v >> 16; };
    auto sw = [](uint32_t v) { return v >> 16 | v << 16; };
    auto m3 = [&](bool mn, uint32_t x, uint32_t y, uint32_t z) {
        auto f = [&](uint32_t p, uint32_t q, uint32_t r) { return mn ? std::min(p, std::min(q, r)) : std::max(p, std::max(q, r)); };
        return f(lo(x), lo(y), lo(z)) | f(hi(x), hi(y), hi(z)) << 16;
    };
    long bad[6] = {};
    for (int i = 0; i < n; i++) {
        const uint32_t e[6] = {m3(true, a[i], b[i], c[i]), m3(true, a[i], b[i], sw(c[i])), m3(true, a[i], sw(b[i]), sw(c[i])),
                               m3(false, a[i], b[i], c[i]), m3(false, a[i], b[i], sw(c[i])), m3(false, a[i], sw(b[i]), sw(c[i]))};
        for (int j = 0; j < 6; j++)
            if (o[6 * i + j] != e[j]) {
                if (bad[j]++ < 3) printf("form %d: a %08x b %08x c %08x -> %08x, expected %08x\n", j, a[i], b[i], c[i], o[6 * i + j], e[j]);
            }
    }
    for (int j = 0; j < 6; j++) printf("form %d: %ld of %d wrong\n", j, bad[j], n);
    return 0;
}
