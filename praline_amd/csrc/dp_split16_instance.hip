// dp_split16_instance.hip -- k_dp_split16 / k_split_f16 / k_scores_tile16 instances; compiled with
// -mllvm -amdgpu-mfma-vgpr-form.
#define PRALINE_SPLIT16_AUX 1
#include "dp_launch.hip.h"
#include "dp_split16.hip.h"
#include "dp_split16_tb.hip.h"

#include <cstdlib>

template <int NR, int NTERM> static void launch16(const LaunchArgs &la, const Arena16Dev &a16, bool local, unsigned wpb)
{
    const dim3 grid((la.n_tasks + wpb - 1) / wpb), block(64 * wpb);
#define PRALINE_LAUNCH16(LOC, BSRC, GRID, BLOCK)                                                                         \
    hipLaunchKernelGGL((k_dp_split16<NR, NTERM, LOC, BSRC>), GRID, BLOCK, 0, la.stream, a16, la.tasks, la.lane_one,       \
                       la.lane_pair, (float2 *)la.bnd, la.scores, la.rp, (int)la.n_tasks)
    if constexpr (NTERM == 1) {
        if (a16.sym8 != nullptr && la.wg != nullptr) {   // one-hot arena, small batch: shared waves + match-score lookup
            const dim3 g4(la.n_wg), b4(256);
            if (local)
                hipLaunchKernelGGL((k_dp_split16<NR, NTERM, true, 3, 4>), g4, b4, 0, la.stream, a16, la.tasks, la.lane_one,
                                   la.lane_pair, (float2 *)la.bnd, la.scores, la.rp, (int)la.n_tasks, la.wg);
            else
                hipLaunchKernelGGL((k_dp_split16<NR, NTERM, false, 3, 4>), g4, b4, 0, la.stream, a16, la.tasks, la.lane_one,
                                   la.lane_pair, (float2 *)la.bnd, la.scores, la.rp, (int)la.n_tasks, la.wg);
            return;
        }
        if (a16.sym8 != nullptr) {
            // one-hot arena: the match scores are looked up (BSRC = 3: no MFMA at all); PRALINE_NO_LOOKUP=1 keeps the
            // one-hot operand table feeding the MFMAs (BSRC = 1)
            const char *nl = getenv("PRALINE_NO_LOOKUP");
            if (!(nl && nl[0] == '1')) {
                const dim3 g1((unsigned)la.n_tasks), b1(64);   // one wave (= its own table) per workgroup
                if (local) PRALINE_LAUNCH16(true, 3, g1, b1); else PRALINE_LAUNCH16(false, 3, g1, b1);
                return;
            }
            if (local) PRALINE_LAUNCH16(true, 1, grid, block); else PRALINE_LAUNCH16(false, 1, grid, block);
            return;
        }
    }
    if (a16.stage && la.wg != nullptr) {  // small batch: four-wave workgroups (see k_dp_split16 WPG)
        const dim3 g4(la.n_wg), b4(256);
        if (local)
            hipLaunchKernelGGL((k_dp_split16<NR, NTERM, true, 2, 4>), g4, b4, 0, la.stream, a16, la.tasks, la.lane_one,
                               la.lane_pair, (float2 *)la.bnd, la.scores, la.rp, (int)la.n_tasks, la.wg);
        else
            hipLaunchKernelGGL((k_dp_split16<NR, NTERM, false, 2, 4>), g4, b4, 0, la.stream, a16, la.tasks, la.lane_one,
                               la.lane_pair, (float2 *)la.bnd, la.scores, la.rp, (int)la.n_tasks, la.wg);
        return;
    }
    if (a16.stage) {  // LDS-staged operand stream: one wave (= its own LDS rings) per workgroup
        const dim3 g1((unsigned)la.n_tasks), b1(64);
        if (local) PRALINE_LAUNCH16(true, 2, g1, b1); else PRALINE_LAUNCH16(false, 2, g1, b1);
        return;
    }
    if (local) PRALINE_LAUNCH16(true, 0, grid, block); else PRALINE_LAUNCH16(false, 0, grid, block);
#undef PRALINE_LAUNCH16
}

int praline_launch_split16(const LaunchArgs &la, const Arena16Dev &a16, int nr, int nterm, bool local)
{
    unsigned wpb = 1;  // wavefronts (= tasks) per workgroup
    if (const char *env = getenv("PRALINE_WPB")) { const int v = atoi(env); if (v >= 1 && v <= 4) wpb = (unsigned)v; }
    if (nr == 1 && nterm == 1) launch16<1, 1>(la, a16, local, wpb);
    else if (nr == 1 && nterm == 3) launch16<1, 3>(la, a16, local, wpb);
    else if (nr == 2 && nterm == 1) launch16<2, 1>(la, a16, local, wpb);
    else if (nr == 2 && nterm == 3) launch16<2, 3>(la, a16, local, wpb);
    else if (nr == 2 && nterm == 2) launch16<2, 2>(la, a16, local, wpb);
    else return PRALINE_ERR_UNSUPPORTED;
    return PRALINE_OK;
}

void praline_launch_split_f16(const float *src, int KP, int KS, int n_active, int NR, int64_t rows_pad, void *dst, int *flag,
                              hipStream_t stream)
{
    const int64_t total = rows_pad * 2 * NR * 8;
    hipLaunchKernelGGL(k_split_f16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, KP, KS, n_active, NR,
                       rows_pad, (_Float16 *)dst, flag);
}

int praline_launch_scores_tile16(const Arena16Dev &a16, int nr, int nterm, int one, int two, int L1, int L2, float *m,
                                 hipStream_t stream)
{
    const dim3 grid((unsigned)((L2 + 31) / 32), (unsigned)((L1 + 31) / 32));
    if (nr == 1 && nterm == 1) hipLaunchKernelGGL((k_scores_tile16<1, 1>), grid, dim3(64), 0, stream, a16, one, two, m);
    else if (nr == 1 && nterm == 3) hipLaunchKernelGGL((k_scores_tile16<1, 3>), grid, dim3(64), 0, stream, a16, one, two, m);
    else if (nr == 2 && nterm == 1) hipLaunchKernelGGL((k_scores_tile16<2, 1>), grid, dim3(64), 0, stream, a16, one, two, m);
    else if (nr == 2 && nterm == 3) hipLaunchKernelGGL((k_scores_tile16<2, 3>), grid, dim3(64), 0, stream, a16, one, two, m);
    else if (nr == 2 && nterm == 2) hipLaunchKernelGGL((k_scores_tile16<2, 2>), grid, dim3(64), 0, stream, a16, one, two, m);
    else return PRALINE_ERR_UNSUPPORTED;
    return PRALINE_OK;
}

template <int NR, int NTERM> static void launch16_tb(const LaunchArgs &la, const Arena16Dev &a16, bool local, bool mask, unsigned wpb)
{
    const dim3 grid((la.n_tasks + wpb - 1) / wpb), block(64 * wpb);
#define PRALINE_TB_LAUNCH(LOC, MSK)                                                                                      \
    hipLaunchKernelGGL((k_dp_split16_tb<NR, NTERM, LOC, MSK>), grid, block, 0, la.stream, a16, la.tasks, la.lane_one,      \
                       la.lane_pair, (float4 *)la.bnd, (uint2 *)la.tb, la.aux, la.rl, la.scores, la.end_cells, la.rp,      \
                       (int)la.n_tasks)
    if constexpr (NTERM == 1) {
        // integer scoring on a one-hot arena: the match scores are looked up (no MFMA, two waves per SIMD);
        // PRALINE_NO_LOOKUP=1 keeps the MFMA instances
        const char *nl = getenv("PRALINE_NO_LOOKUP");
        if (a16.sym8 != nullptr && !(nl && nl[0] == '1')) {
#define PRALINE_TB_LAUNCH_LK(LOC, MSK)                                                                                   \
    hipLaunchKernelGGL((k_dp_split16_tb<NR, NTERM, LOC, MSK, false, false, 3>), grid, block, 0, la.stream, a16, la.tasks,  \
                       la.lane_one, la.lane_pair, (float4 *)la.bnd, (uint2 *)la.tb, la.aux, la.rl, la.scores, la.end_cells,  \
                       la.rp, (int)la.n_tasks)
            if (local) { if (mask) PRALINE_TB_LAUNCH_LK(true, true); else PRALINE_TB_LAUNCH_LK(true, false); }
            else { if (mask) PRALINE_TB_LAUNCH_LK(false, true); else PRALINE_TB_LAUNCH_LK(false, false); }
#undef PRALINE_TB_LAUNCH_LK
            return;
        }
    }
    if (local) { if (mask) PRALINE_TB_LAUNCH(true, true); else PRALINE_TB_LAUNCH(true, false); }
    else { if (mask) PRALINE_TB_LAUNCH(false, true); else PRALINE_TB_LAUNCH(false, false); }
#undef PRALINE_TB_LAUNCH
}

// chain mode (dp_split16_tb.hip.h): one wave per (task, strip), grid strip-major
template <int NR, int NTERM> static void launch16_tb_chain(const LaunchArgs &la, const Arena16Dev &a16, bool local, bool mask,
                                                            int max_strips, int *flags, void *cand, int every)
{
    const dim3 grid(la.n_tasks * (unsigned)max_strips), block(64);
#define PRALINE_CHAIN_LAUNCH(LOC, MSK)                                                                                  \
    hipLaunchKernelGGL((k_dp_split16_tb<NR, NTERM, LOC, MSK, true>), grid, block, 0, la.stream, a16, la.tasks, la.lane_one, \
                       la.lane_pair, (float4 *)la.bnd, (uint2 *)la.tb, la.aux, la.rl, la.scores, la.end_cells, la.rp,       \
                       (int)la.n_tasks, flags, max_strips + 1, (float4 *)cand, every)
    if constexpr (NTERM == 1) {
        const char *nl = getenv("PRALINE_NO_LOOKUP");
        if (a16.sym8 != nullptr && !(nl && nl[0] == '1')) {
#define PRALINE_CHAIN_LAUNCH_LK(LOC, MSK)                                                                                \
    hipLaunchKernelGGL((k_dp_split16_tb<NR, NTERM, LOC, MSK, true, false, 3>), grid, block, 0, la.stream, a16, la.tasks,   \
                       la.lane_one, la.lane_pair, (float4 *)la.bnd, (uint2 *)la.tb, la.aux, la.rl, la.scores, la.end_cells,  \
                       la.rp, (int)la.n_tasks, flags, max_strips + 1, (float4 *)cand, every)
            if (local) { if (mask) PRALINE_CHAIN_LAUNCH_LK(true, true); else PRALINE_CHAIN_LAUNCH_LK(true, false); }
            else { if (mask) PRALINE_CHAIN_LAUNCH_LK(false, true); else PRALINE_CHAIN_LAUNCH_LK(false, false); }
#undef PRALINE_CHAIN_LAUNCH_LK
            return;
        }
    }
    if (local) { if (mask) PRALINE_CHAIN_LAUNCH(true, true); else PRALINE_CHAIN_LAUNCH(true, false); }
    else { if (mask) PRALINE_CHAIN_LAUNCH(false, true); else PRALINE_CHAIN_LAUNCH(false, false); }
#undef PRALINE_CHAIN_LAUNCH
}

int praline_launch_split16_tb_chain(const LaunchArgs &la, const Arena16Dev &a16, int nr, int nterm, bool local, bool mask,
                                    int max_strips, int *flags, void *cand, int every)
{
    if (nr == 1 && nterm == 1) launch16_tb_chain<1, 1>(la, a16, local, mask, max_strips, flags, cand, every);
    else if (nr == 1 && nterm == 3) launch16_tb_chain<1, 3>(la, a16, local, mask, max_strips, flags, cand, every);
    else if (nr == 2 && nterm == 1) launch16_tb_chain<2, 1>(la, a16, local, mask, max_strips, flags, cand, every);
    else if (nr == 2 && nterm == 3) launch16_tb_chain<2, 3>(la, a16, local, mask, max_strips, flags, cand, every);
    else if (nr == 2 && nterm == 2) launch16_tb_chain<2, 2>(la, a16, local, mask, max_strips, flags, cand, every);
    else return PRALINE_ERR_UNSUPPORTED;
    return PRALINE_OK;
}

int praline_launch_split16_tb(const LaunchArgs &la, const Arena16Dev &a16, int nr, int nterm, bool local, bool mask)
{
    unsigned wpb = 1;
    if (const char *env = getenv("PRALINE_WPB")) { const int v = atoi(env); if (v >= 1 && v <= 4) wpb = (unsigned)v; }
    if (nr == 1 && nterm == 1) launch16_tb<1, 1>(la, a16, local, mask, wpb);
    else if (nr == 1 && nterm == 3) launch16_tb<1, 3>(la, a16, local, mask, wpb);
    else if (nr == 2 && nterm == 1) launch16_tb<2, 1>(la, a16, local, mask, wpb);
    else if (nr == 2 && nterm == 3) launch16_tb<2, 3>(la, a16, local, mask, wpb);
    else if (nr == 2 && nterm == 2) launch16_tb<2, 2>(la, a16, local, mask, wpb);
    else return PRALINE_ERR_UNSUPPORTED;
    return PRALINE_OK;
}

#ifdef PRALINE_TRACE
extern "C" int praline_trace_set(void *device_buffer)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(praline_trace_buf), &device_buffer, sizeof(void *)) == hipSuccess ? 0 : -2;
}
#endif
