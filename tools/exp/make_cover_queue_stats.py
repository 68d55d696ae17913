#!/usr/bin/env python3
"""Writes tools/exp/cover_queue_stats.hip: filtlong_amd/csrc/cover_queue.hip with counters in the kernel that gives every lane a diagonal of
its own (lanes that find one, lanes nothing is known about, pieces queued, exact-table loads, pieces in which a lookup finds a member the
text did not know).  Build it into a library of its own and run tools/exp/cq_stats.py with FLX_LIB_PATH pointing at it
(tools/gpu_calls/r06_call31.sh).  Result on synthetic profile 1, 10^6 reads (profiles/r06_microbench.txt §8)."""
import os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(R, "filtlong_amd", "csrc", "cover_queue.hip")).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, old[:60]
    s = s.replace(old, new)


s = s.replace('namespace {\n', '__device__ unsigned long long flx_cq_stats_d[32];\nextern "C" int flx_debug_cq_stats(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(flx_cq_stats_d), sizeof(unsigned long long) * 32); }\nextern "C" int flx_debug_cq_stats_reset() { unsigned long long z[32] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(flx_cq_stats_d), z, sizeof z); }\n#define ST(i, v) do { const unsigned long long v_ = (v); if (lane == 0 && v_) atomicAdd(&flx_cq_stats_d[i], v_); } while (0)\n#define STB(i, pred) ST(i, (unsigned long long)__popcll(__ballot(pred)))\nnamespace {\n', 1)
old = '''                    const int dl = ld.dl;
                    const bool matched = ld.matched != 0;
'''
rep(old, old + '''                    ST(0, 1); STB(1, (valid16 >> 15) != 0); STB(2, matched); STB(3, matched && dl != 0); STB(4, (valid16 >> 15) != 0 && !matched && known == 0);
                    STB(5, known != 0); ST(6, (unsigned long long)(dl < 0 ? -dl : dl) >= 12 && matched ? 1 : 0);
''')
rep('''            S.ring[sp & (kRing - 1)][lane] = (uint16_t)known;
            const unsigned long long nb = __ballot(need);''', '''            S.ring[sp & (kRing - 1)][lane] = (uint16_t)known;
            const unsigned long long nb = __ballot(need);
            if (INDELS) { ST(8, 1); ST(9, (unsigned long long)__popcll(nb)); STB(10, valid16 != 0); }''')
old = '''            auto probe = [&](int top, int bot) {  // positions asked from above / from below, -1 = none
'''
rep(old, old + '''                if (INDELS) { ST(11, (unsigned long long)__popcll(__ballot(top >= 0)) + __popcll(__ballot(bot >= 0))); }
''')
rep('''            if (act) S.ring[(id >> 6) & (kRing - 1)][id & 63u] = (uint16_t)hits;''', '''            if (INDELS) { STB(12, act && (hits & ~known) != 0); ST(13, 1); STB(14, act); }
            if (act) S.ring[(id >> 6) & (kRing - 1)][id & 63u] = (uint16_t)hits;''')
open(os.path.join(R, "tools", "exp", "cover_queue_stats.hip"), "w").write(s)
print("wrote tools/exp/cover_queue_stats.hip")
