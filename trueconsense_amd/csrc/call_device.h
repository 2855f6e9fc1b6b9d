// call_device.h — the per-position call as a device function, shared by call.hip (one lane per position
// of a finished matrix) and tally_common.h (ride-along call: the first blocks of a tally launch call the matrix an
// earlier launch finished).
// What it replaces in the reference is listed in call.hip.
#pragma once
#include "tcmi_internal.h"

namespace tcmi_calldev {

__device__ inline void cswap(int64_t &a, int64_t &b)
{   // descending
    const int64_t hi = a > b ? a : b, lo = a > b ? b : a;
    a = hi; b = lo;
}

__device__ inline char iupac_of(unsigned mask)
{   // bit0 A, bit1 C, bit2 G, bit3 T -> "?ACMGRSV" "TWYHKDBN" packed little-endian (Ambig.py:1-15 tables)
    const uint64_t tab = (mask & 8u) ? 0x4E42444B48595754ull : 0x565352474D43413Full;
    return (char)((tab >> (8u * (mask & 7u))) & 0xFFu);
}

__device__ inline char letter_of(unsigned rank)
{   // "ACGTX" packed little-endian
    return (char)((0x5854474341ull >> (8u * rank)) & 0xFFu);
}

struct Record { uint8_t plain, alt, flags; };

// counters of one position -> its call record (Sequences.py:119-165, 229-235; Ambig.py:18-228; Events.py:29-36, 85-106)
__device__ inline Record call_position(int64_t cov, int64_t nA, int64_t nT, int64_t nC, int64_t nG, int64_t nX, int64_t nI,
                                       int32_t mincov, int include_ambig)
{
    unsigned f = 0;
    // key = count*8 + letter rank (A<C<G<T<X): Python's sort of (count, letter) tuples
    int64_t k0 = nA * 8 + 0, k1 = nC * 8 + 1, k2 = nG * 8 + 2, k3 = nT * 8 + 3, k4 = nX * 8 + 4;
    // 9-comparator sorting network for 5 keys, descending
    cswap(k0, k1); cswap(k3, k4); cswap(k2, k4); cswap(k2, k3); cswap(k0, k3);
    cswap(k0, k2); cswap(k1, k4); cswap(k1, k3); cswap(k1, k2);
    const unsigned r1 = (unsigned)(k0 & 7), r2 = (unsigned)(k1 & 7), r3 = (unsigned)(k2 & 7);
    const int64_t c1 = k0 >> 3, c2 = k1 >> 3, c3 = k2 >> 3, c4 = k3 >> 3;

    if (cov < mincov) f |= TCMI_F_LOWCOV;
    if (r1 == 4) f |= TCMI_F_PRIMX;
    if (cov > mincov) f |= TCMI_F_COVGT;
    const double dcov = (double)cov;
    if (cov == 0) f |= TCMI_F_COVZERO;
    else if (((double)nX / dcov) * 100.0 >= 15.0) f |= TCMI_F_MINDEL;
    if (cov >= mincov && cov != 0 && nI != 0 && ((double)nI / dcov) * 100.0 > 55.0) f |= TCMI_F_INSCAND;

    char amb = 0;
    if (cov != 0 && r1 != 4 && r2 != 4) {
        const double p1 = ((double)c1 / dcov) * 100.0, p2 = ((double)c2 / dcov) * 100.0;
        const double p3 = ((double)c3 / dcov) * 100.0, p4 = ((double)c4 / dcov) * 100.0;
        if (fabs(p1 - p2) <= 10.0) {
            const unsigned m2 = (1u << r1) | (1u << r2);
            if (fabs(p1 - p3) <= 10.0 && fabs(p2 - p3) <= 10.0) {
                if (fabs(p1 - p4) <= 10.0 && fabs(p2 - p4) <= 10.0 && fabs(p3 - p4) <= 10.0) amb = 'N';
                else if (r3 == 4) amb = 'N';
                else amb = iupac_of(m2 | (1u << r3));
            } else amb = iupac_of(m2);
        }
    }
    if (amb) f |= TCMI_F_AMBIG;
    char ch1 = letter_of(r1), ch2 = letter_of(r2);
    if (c1 < mincov) ch1 = (char)(ch1 | 0x20);
    if (c2 < mincov) ch2 = (char)(ch2 | 0x20);
    Record r;
    r.plain = (uint8_t)((f & TCMI_F_LOWCOV) ? 'N' : ((include_ambig && amb) ? amb : ch1));
    r.alt = (uint8_t)ch2;
    r.flags = (uint8_t)f;
    return r;
}

} // namespace tcmi_calldev
