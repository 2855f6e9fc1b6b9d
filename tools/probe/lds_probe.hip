// LDS cost of the match loop's read / write pair by alignment and width, 16 wavefronts a CU (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define LOOP(RD, WR, MR, MW, DR) \
        asm volatile( \
            "s_mov_b64 s[92:93], exec\n" \
            "s_mov_b64 s[82:83], %[mm]\n" \
            "s_ff1_i32_b64 s84, s[82:83]\n" \
            "v_readlane_b32 %[sa], %[vA2], s84\n" \
            "v_readlane_b32 %[sb], %[vB2], s84\n" \
            "s_bitset0_b64 s[82:83], s84\n" \
            "LMa%=:\n" \
            "v_cmpx_ge_u32 vcc, %[sa], %[vX8]\n" \
            "v_min_u32_sdwa %[t2], %[sa], %[vlane8] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
            "v_add_u32 %[t0], %[sb], %[t2]\n" \
            "v_add_u32_sdwa %[t1], %[sa], %[t2] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n" \
            "v_and_b32 %[t0], " MR ", %[t0]\n" \
            "v_and_b32 %[t1], " MW ", %[t1]\n" \
            RD " " DR ", %[t0]\n" \
            "s_ff1_i32_b64 s84, s[82:83]\n" \
            "s_cmp_eq_u64 s[82:83], 0\n" \
            "v_readlane_b32 %[sa], %[vA2], s84\n" \
            "v_readlane_b32 %[sb], %[vB2], s84\n" \
            "s_bitset0_b64 s[82:83], s84\n" \
            "s_waitcnt lgkmcnt(0)\n" \
            WR " %[t1], " DR "\n" \
            "s_mov_b64 exec, s[92:93]\n" \
            "s_cbranch_scc0 LMa%=\n" \
            : [sa] "=&s"(sa), [sb] "=&s"(sb), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t23] "=&v"(t23) \
            : [mm] "s"(mm), [vA2] "v"(vA2), [vB2] "v"(vB2), [vX8] "v"(lane8_hi), [vlane8] "v"(lane8) \
            : "s82", "s83", "s84", "s92", "s93", "vcc", "scc", "memory");
template <int V, int FULL>
__global__ __launch_bounds__(64) void probe(unsigned long long *out, int reps)
{
    __shared__ unsigned char ring[8192 + 576];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8192 + 576; i += 64) ring[i] = (unsigned char)i;
    __syncthreads();
    const unsigned len = FULL ? 512u : 40u + (unsigned)(lane * 37 % 200);       // FULL: every lane has a piece
    const unsigned dm = 600u + (unsigned)lane * 101u, sm = dm - 301u;
    const unsigned vA2 = ((len - 1u) << 16) | ((dm - 7u) & 0xFFFFu), vB2 = sm - 7u;
    const unsigned lane8 = (unsigned)lane * 8u + 7u, lane8_hi = (unsigned)lane << 19;
    unsigned sa, sb, t0, t1, t2; unsigned long long t23;
    const unsigned long long t_a = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        unsigned long long mm = ~0ull;
        if (V == 0) LOOP("ds_read_b64", "ds_write_b64", "0xffff", "0xffff", "%[t23]")
        if (V == 1) LOOP("ds_read_b64", "ds_write_b64", "0xffff", "0xfff8", "%[t23]")
        if (V == 2) LOOP("ds_read_b64", "ds_write_b64", "0xfff8", "0xffff", "%[t23]")
        if (V == 3) LOOP("ds_read_b64", "ds_write_b64", "0xfff8", "0xfff8", "%[t23]")
        if (V == 4) LOOP("ds_read_b64", "ds_write_b64", "0xfffc", "0xfffc", "%[t23]")
        if (V == 5) LOOP("ds_read_b32", "ds_write_b32", "0xffff", "0xffff", "%[t2]")
        if (V == 6) LOOP("ds_read_b32", "ds_write_b32", "0xffff", "0xfffc", "%[t2]")
        if (V == 7) LOOP("ds_read_b32", "ds_write_b32", "0xfffc", "0xfffc", "%[t2]")
        if (V == 8) LOOP("ds_read_u8", "ds_write_b8", "0xffff", "0xffff", "%[t2]")
        if (V == 9) LOOP("ds_read_u16", "ds_write_b16", "0xffff", "0xffff", "%[t2]")
    }
    const unsigned long long t_b = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x] = t_b - t_a;
    if (lane == 63 && ring[17] == 255 && t0 == 12345 && t1 == 999 && (unsigned)t23 == 77 && t2 == 5) out[0] = 0;
}
template <int V, int FULL = 0> static void run(const char *name, int blocks, int reps)
{
    unsigned long long *d; hipMalloc(&d, blocks * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<V, FULL><<<blocks, 64>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(a); probe<V, FULL><<<blocks, 64>>>(d, reps); hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += (double)x;
    std::printf("%-44s blocks %5d: %7.1f ticks per match per wavefront, %.1f ns per match\n", name, blocks, s / blocks / reps / 64, ms * 1e6 / reps / 64);
    hipFree(d);
}
int main()
{
    const int reps = 200;
    for (int blocks : {256, 4096}) {
        run<0>("b64: read any, write any", blocks, reps);
        run<1>("b64: read any, write 8-aligned", blocks, reps);
        run<2>("b64: read 8-aligned, write any", blocks, reps);
        run<3>("b64: both 8-aligned", blocks, reps);
        run<4>("b64: both 4-aligned", blocks, reps);
        run<5>("b32: both any", blocks, reps);
        run<6>("b32: read any, write aligned", blocks, reps);
        run<7>("b32: both aligned", blocks, reps);
        run<8>("bytes", blocks, reps);
        run<9>("b16 any", blocks, reps);
        run<0, 1>("b64 any / any, all 64 lanes", blocks, reps);
        run<3, 1>("b64 8-aligned, all 64 lanes", blocks, reps);
        run<7, 1>("b32 aligned, all 64 lanes", blocks, reps);
        run<8, 1>("bytes, all 64 lanes", blocks, reps);
    }
    return 0;
}
