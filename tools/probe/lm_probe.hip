// cycles per iteration of bgzf_copy's plain-match loop under its real residency (diagnostic; hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
template <int V>
__global__ __launch_bounds__(64) void probe(unsigned long long *out, int reps)
{
    __shared__ unsigned char ring[8192 + 576];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8192 + 576; i += 64) ring[i] = (unsigned char)i;
    __syncthreads();
    const unsigned len = 40u + (unsigned)(lane * 37 % 200);            // 40 .. 239
    const unsigned dm = 600u + (unsigned)lane * 100u, sm = dm - 300u;
    const unsigned vA2 = ((len - 1u) << 16) | ((dm - 7u) & 0xFFFFu), vB2 = sm - 7u;
    const unsigned lane8 = (unsigned)lane * 8u + 7u, lane8_hi = (unsigned)lane << 19;
    unsigned sa, sb, t0, t1, t2; unsigned long long t23;
    const unsigned long long t_a = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        unsigned long long mm = ~0ull;
        if (V == 0)
        asm volatile(
            "s_mov_b64 s[92:93], exec\n"
            "s_mov_b64 s[82:83], %[mm]\n"
            "s_ff1_i32_b64 s84, s[82:83]\n"
            "v_readlane_b32 %[sa], %[vA2], s84\n"
            "v_readlane_b32 %[sb], %[vB2], s84\n"
            "s_bitset0_b64 s[82:83], s84\n"
            "LMa%=:\n"
            "v_cmpx_ge_u32 vcc, %[sa], %[vX8]\n"
            "v_min_u32_sdwa %[t2], %[sa], %[vlane8] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
            "v_add_u32 %[t0], %[sb], %[t2]\n"
            "v_add_u32_sdwa %[t1], %[sa], %[t2] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
            "ds_read_b64 %[t23], %[t0]\n"
            "s_ff1_i32_b64 s84, s[82:83]\n"
            "s_cmp_eq_u64 s[82:83], 0\n"
            "v_readlane_b32 %[sa], %[vA2], s84\n"
            "v_readlane_b32 %[sb], %[vB2], s84\n"
            "s_bitset0_b64 s[82:83], s84\n"
            "s_waitcnt lgkmcnt(0)\n"
            "ds_write_b64 %[t1], %[t23]\n"
            "s_mov_b64 exec, s[92:93]\n"
            "s_cbranch_scc0 LMa%=\n"
            : [sa] "=&s"(sa), [sb] "=&s"(sb), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t23] "=&v"(t23)
            : [mm] "s"(mm), [vA2] "v"(vA2), [vB2] "v"(vB2), [vX8] "v"(lane8_hi), [vlane8] "v"(lane8)
            : "s82", "s83", "s84", "s92", "s93", "vcc", "scc", "memory");
        if (V == 1)                                             // no LDS
        asm volatile(
            "s_mov_b64 s[92:93], exec\n"
            "s_mov_b64 s[82:83], %[mm]\n"
            "s_ff1_i32_b64 s84, s[82:83]\n"
            "v_readlane_b32 %[sa], %[vA2], s84\n"
            "v_readlane_b32 %[sb], %[vB2], s84\n"
            "s_bitset0_b64 s[82:83], s84\n"
            "LMa%=:\n"
            "v_cmpx_ge_u32 vcc, %[sa], %[vX8]\n"
            "v_min_u32_sdwa %[t2], %[sa], %[vlane8] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
            "v_add_u32 %[t0], %[sb], %[t2]\n"
            "v_add_u32_sdwa %[t1], %[sa], %[t2] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
            "s_ff1_i32_b64 s84, s[82:83]\n"
            "s_cmp_eq_u64 s[82:83], 0\n"
            "v_readlane_b32 %[sa], %[vA2], s84\n"
            "v_readlane_b32 %[sb], %[vB2], s84\n"
            "s_bitset0_b64 s[82:83], s84\n"
            "s_mov_b64 exec, s[92:93]\n"
            "s_cbranch_scc0 LMa%=\n"
            : [sa] "=&s"(sa), [sb] "=&s"(sb), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t23] "=&v"(t23)
            : [mm] "s"(mm), [vA2] "v"(vA2), [vB2] "v"(vB2), [vX8] "v"(lane8_hi), [vlane8] "v"(lane8)
            : "s82", "s83", "s84", "s92", "s93", "vcc", "scc", "memory");
        if (V == 2)                                             // LDS only: fixed operands, no readlanes
        asm volatile(
            "s_mov_b64 s[92:93], exec\n"
            "s_mov_b64 s[82:83], %[mm]\n"
            "v_readlane_b32 %[sa], %[vA2], 5\n"
            "v_readlane_b32 %[sb], %[vB2], 5\n"
            "LMa%=:\n"
            "s_ff1_i32_b64 s84, s[82:83]\n"
            "s_bitset0_b64 s[82:83], s84\n"
            "v_cmpx_ge_u32 vcc, %[sa], %[vX8]\n"
            "v_min_u32_sdwa %[t2], %[sa], %[vlane8] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
            "v_add_u32 %[t0], %[sb], %[t2]\n"
            "v_add_u32_sdwa %[t1], %[sa], %[t2] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
            "ds_read_b64 %[t23], %[t0]\n"
            "s_cmp_eq_u64 s[82:83], 0\n"
            "s_waitcnt lgkmcnt(0)\n"
            "ds_write_b64 %[t1], %[t23]\n"
            "s_mov_b64 exec, s[92:93]\n"
            "s_cbranch_scc0 LMa%=\n"
            : [sa] "=&s"(sa), [sb] "=&s"(sb), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t23] "=&v"(t23)
            : [mm] "s"(mm), [vA2] "v"(vA2), [vB2] "v"(vB2), [vX8] "v"(lane8_hi), [vlane8] "v"(lane8)
            : "s82", "s83", "s84", "s92", "s93", "vcc", "scc", "memory");
        if (V == 3)                                             // aligned 4-byte LDS ops instead of unaligned 8-byte ones
        asm volatile(
            "s_mov_b64 s[92:93], exec\n"
            "s_mov_b64 s[82:83], %[mm]\n"
            "s_ff1_i32_b64 s84, s[82:83]\n"
            "v_readlane_b32 %[sa], %[vA2], s84\n"
            "v_readlane_b32 %[sb], %[vB2], s84\n"
            "s_bitset0_b64 s[82:83], s84\n"
            "LMa%=:\n"
            "v_cmpx_ge_u32 vcc, %[sa], %[vX8]\n"
            "v_min_u32_sdwa %[t2], %[sa], %[vlane8] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
            "v_add_u32 %[t0], %[sb], %[t2]\n"
            "v_add_u32_sdwa %[t1], %[sa], %[t2] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
            "v_and_b32 %[t0], 0xfffc, %[t0]\n"
            "v_and_b32 %[t1], 0xfffc, %[t1]\n"
            "ds_read_b32 %[t2], %[t0]\n"
            "s_ff1_i32_b64 s84, s[82:83]\n"
            "s_cmp_eq_u64 s[82:83], 0\n"
            "v_readlane_b32 %[sa], %[vA2], s84\n"
            "v_readlane_b32 %[sb], %[vB2], s84\n"
            "s_bitset0_b64 s[82:83], s84\n"
            "s_waitcnt lgkmcnt(0)\n"
            "ds_write_b32 %[t1], %[t2]\n"
            "s_mov_b64 exec, s[92:93]\n"
            "s_cbranch_scc0 LMa%=\n"
            : [sa] "=&s"(sa), [sb] "=&s"(sb), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t23] "=&v"(t23)
            : [mm] "s"(mm), [vA2] "v"(vA2), [vB2] "v"(vB2), [vX8] "v"(lane8_hi), [vlane8] "v"(lane8)
            : "s82", "s83", "s84", "s92", "s93", "vcc", "scc", "memory");
        // ---- round 4's byte rounds (a byte a lane, matches of up to 64 bytes): as shipped (4), two matches in flight (5), without LDS (6)
        const unsigned len1 = 8u + (unsigned)(lane * 37 % 56);          // 8 .. 63
        const unsigned vA1 = ((len1 - 1u) << 16) | ((dm - 7u) & 0xFFFFu);
        const unsigned lane_sh16 = (unsigned)lane << 16, lane_p7 = (unsigned)lane + 7u;
        unsigned u0, u1, u2;
        if (V == 4)
        asm volatile(
            "s_mov_b64 s[92:93], exec\n"
            "s_mov_b64 s[82:83], %[mm]\n"
            "LMb%=:\n"
            "s_ff1_i32_b64 s84, s[82:83]\n"
            "v_readlane_b32 %[sa], %[vA2], s84\n"
            "v_readlane_b32 %[sb], %[vB2], s84\n"
            "s_bitset0_b64 s[82:83], s84\n"
            "s_cmp_lt_u32 %[sa], 0x400000\n"
            "s_cbranch_scc0 LXb%=\n"
            "v_cmpx_ge_u32 vcc, %[sa], %[vX1]\n"
            "v_add_u32 %[t0], %[sb], %[vlane7]\n"
            "v_add_u32_sdwa %[t1], %[sa], %[vlane7] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
            "ds_read_u8 %[t2], %[t0]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "ds_write_b8 %[t1], %[t2]\n"
            "s_mov_b64 exec, s[92:93]\n"
            "s_cmp_lg_u64 s[82:83], 0\n"
            "s_cbranch_scc1 LMb%=\n"
            "LXb%=:\n"
            "s_mov_b64 exec, s[92:93]\n"
            : [sa] "=&s"(sa), [sb] "=&s"(sb), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
            : [mm] "s"(mm), [vA2] "v"(vA1), [vB2] "v"(vB2), [vX1] "v"(lane_sh16), [vlane7] "v"(lane_p7)
            : "s82", "s83", "s84", "s92", "s93", "vcc", "scc", "memory");
        if (V == 6)
        asm volatile(
            "s_mov_b64 s[92:93], exec\n"
            "s_mov_b64 s[82:83], %[mm]\n"
            "LMb%=:\n"
            "s_ff1_i32_b64 s84, s[82:83]\n"
            "v_readlane_b32 %[sa], %[vA2], s84\n"
            "v_readlane_b32 %[sb], %[vB2], s84\n"
            "s_bitset0_b64 s[82:83], s84\n"
            "s_cmp_lt_u32 %[sa], 0x400000\n"
            "s_cbranch_scc0 LXb%=\n"
            "v_cmpx_ge_u32 vcc, %[sa], %[vX1]\n"
            "v_add_u32 %[t0], %[sb], %[vlane7]\n"
            "v_add_u32_sdwa %[t1], %[sa], %[vlane7] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
            "s_mov_b64 exec, s[92:93]\n"
            "s_cmp_lg_u64 s[82:83], 0\n"
            "s_cbranch_scc1 LMb%=\n"
            "LXb%=:\n"
            "s_mov_b64 exec, s[92:93]\n"
            : [sa] "=&s"(sa), [sb] "=&s"(sb), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
            : [mm] "s"(mm), [vA2] "v"(vA1), [vB2] "v"(vB2), [vX1] "v"(lane_sh16), [vlane7] "v"(lane_p7)
            : "s82", "s83", "s84", "s92", "s93", "vcc", "scc", "memory");
        if (V == 7)                                             // no exec mask: lanes beyond the match copy its last byte once more
        {
        const unsigned vA7 = ((len1 - 1u) << 16) | (dm & 0xFFFFu);
        asm volatile(
            "s_mov_b64 s[82:83], %[mm]\n"
            "LMb%=:\n"
            "s_ff1_i32_b64 s84, s[82:83]\n"
            "v_readlane_b32 %[sa], %[vA2], s84\n"
            "v_readlane_b32 %[sb], %[vB2], s84\n"
            "s_bitset0_b64 s[82:83], s84\n"
            "s_cmp_lt_u32 %[sa], 0x400000\n"
            "s_cbranch_scc0 LXb%=\n"
            "v_min_u32_sdwa %[t2], %[sa], %[vlane] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
            "v_add_u32 %[t0], %[sb], %[t2]\n"
            "v_add_u32_sdwa %[t1], %[sa], %[t2] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
            "ds_read_u8 %[t2], %[t0]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "ds_write_b8 %[t1], %[t2]\n"
            "s_cmp_lg_u64 s[82:83], 0\n"
            "s_cbranch_scc1 LMb%=\n"
            "LXb%=:\n"
            : [sa] "=&s"(sa), [sb] "=&s"(sb), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
            : [mm] "s"(mm), [vA2] "v"(vA7), [vB2] "v"(sm), [vlane] "v"((unsigned)lane)
            : "s82", "s83", "s84", "vcc", "scc", "memory");
        }
        if (V == 5)                                             // two matches in flight: the next one's read goes out before this one's write
        asm volatile(
            "s_mov_b64 s[92:93], exec\n"
            "s_mov_b64 s[82:83], %[mm]\n"
            // prologue: the first match -> set A, its read under way
            "s_ff1_i32_b64 s84, s[82:83]\n"
            "v_readlane_b32 %[sa], %[vA2], s84\n"
            "v_readlane_b32 %[sb], %[vB2], s84\n"
            "s_bitset0_b64 s[82:83], s84\n"
            "v_cmpx_ge_u32 vcc, %[sa], %[vX1]\n"
            "s_mov_b64 s[96:97], exec\n"
            "v_add_u32 %[t0], %[sb], %[vlane7]\n"
            "v_add_u32_sdwa %[t1], %[sa], %[vlane7] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
            "ds_read_u8 %[t2], %[t0]\n"
            "s_mov_b64 exec, s[92:93]\n"
            "LPa%=:\n"                                           // A in flight: fetch B
            "s_cmp_eq_u64 s[82:83], 0\n"
            "s_cbranch_scc1 LDa%=\n"
            "s_ff1_i32_b64 s84, s[82:83]\n"
            "v_readlane_b32 %[sa], %[vA2], s84\n"
            "v_readlane_b32 %[sb], %[vB2], s84\n"
            "s_bitset0_b64 s[82:83], s84\n"
            "s_cmp_lt_u32 %[sa], 0x400000\n"
            "s_cbranch_scc0 LDa%=\n"
            "v_cmpx_ge_u32 vcc, %[sa], %[vX1]\n"
            "s_mov_b64 s[94:95], exec\n"
            "v_add_u32 %[u0], %[sb], %[vlane7]\n"
            "v_add_u32_sdwa %[u1], %[sa], %[vlane7] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
            "ds_read_u8 %[u2], %[u0]\n"
            "s_mov_b64 exec, s[96:97]\n"
            "s_waitcnt lgkmcnt(1)\n"
            "ds_write_b8 %[t1], %[t2]\n"
            "s_mov_b64 exec, s[92:93]\n"
            // B in flight: fetch A
            "s_cmp_eq_u64 s[82:83], 0\n"
            "s_cbranch_scc1 LDb%=\n"
            "s_ff1_i32_b64 s84, s[82:83]\n"
            "v_readlane_b32 %[sa], %[vA2], s84\n"
            "v_readlane_b32 %[sb], %[vB2], s84\n"
            "s_bitset0_b64 s[82:83], s84\n"
            "s_cmp_lt_u32 %[sa], 0x400000\n"
            "s_cbranch_scc0 LDb%=\n"
            "v_cmpx_ge_u32 vcc, %[sa], %[vX1]\n"
            "s_mov_b64 s[96:97], exec\n"
            "v_add_u32 %[t0], %[sb], %[vlane7]\n"
            "v_add_u32_sdwa %[t1], %[sa], %[vlane7] dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
            "ds_read_u8 %[t2], %[t0]\n"
            "s_mov_b64 exec, s[94:95]\n"
            "s_waitcnt lgkmcnt(1)\n"
            "ds_write_b8 %[u1], %[u2]\n"
            "s_mov_b64 exec, s[92:93]\n"
            "s_branch LPa%=\n"
            "LDa%=:\n"                                           // drain A
            "s_mov_b64 exec, s[96:97]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "ds_write_b8 %[t1], %[t2]\n"
            "s_branch LXc%=\n"
            "LDb%=:\n"                                           // drain B
            "s_mov_b64 exec, s[94:95]\n"
            "s_waitcnt lgkmcnt(0)\n"
            "ds_write_b8 %[u1], %[u2]\n"
            "LXc%=:\n"
            "s_mov_b64 exec, s[92:93]\n"
            : [sa] "=&s"(sa), [sb] "=&s"(sb), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [u0] "=&v"(u0), [u1] "=&v"(u1), [u2] "=&v"(u2)
            : [mm] "s"(mm), [vA2] "v"(vA1), [vB2] "v"(vB2), [vX1] "v"(lane_sh16), [vlane7] "v"(lane_p7)
            : "s82", "s83", "s84", "s92", "s93", "s94", "s95", "s96", "s97", "vcc", "scc", "memory");
    }
    const unsigned long long t_b = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x] = t_b - t_a;
    if (lane == 63 && ring[17] == 255 && t0 == 12345 && t1 == 999 && (unsigned)t23 == 77 && t2 == 5) out[0] = 0;   // (keep everything alive)
}
template <int V> static void run(const char *name, int blocks, int reps)
{
    unsigned long long *d; hipMalloc(&d, blocks * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<V><<<blocks, 64>>>(d, 10); hipDeviceSynchronize();
    hipEventRecord(a); probe<V><<<blocks, 64>>>(d, reps); hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += (double)x;
    std::printf("%-34s blocks %5d: %7.1f ticks per match per wavefront, kernel %.1f us, %.1f ns per match\n", name, blocks, s / blocks / reps / 64, ms * 1e3, ms * 1e6 / reps / 64);
    hipFree(d);
}
int main()
{
    const int reps = 200;
    for (int blocks : {256, 1024, 4096}) {
        run<0>("loop as shipped", blocks, reps);
        run<1>("without the LDS read / write", blocks, reps);
        run<2>("LDS only (operands fixed)", blocks, reps);
        run<3>("aligned 4-byte LDS ops", blocks, reps);
        run<4>("round 4: byte rounds as shipped", blocks, reps);
        run<5>("round 4: two matches in flight", blocks, reps);
        run<6>("round 4: byte rounds without LDS", blocks, reps);
        run<7>("byte rounds, no exec mask (min)", blocks, reps);
    }
    return 0;
}
