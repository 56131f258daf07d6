// praline_raw.hip.h -- part of praline_dp.hip (one translation unit; included there, in this order): parity-layout entry points (the drop-in twins of the six native functions of the reference), debug and diagnostic entry points.
// --------------------------------------------------------------------------------------------
// parity-layout entry points (strided host buffers <-> contiguous device copies)
// --------------------------------------------------------------------------------------------
static bool arr_ok(const praline_array *a) { return a && a->data; }

template <typename T> static void gather2(const praline_array &a, std::vector<T> &out)
{
    const int64_t R = a.dim[0], C = a.dim[1];
    out.resize((size_t)(R * C));
    const char *base = (const char *)a.data;
    for (int64_t r = 0; r < R; ++r)
        for (int64_t c = 0; c < C; ++c) out[(size_t)(r * C + c)] = *(const T *)(base + r * a.stride[0] + c * a.stride[1]);
}

template <typename T> static void gather3(const praline_array &a, std::vector<T> &out)
{
    const int64_t R = a.dim[0], C = a.dim[1], K = a.dim[2];
    out.resize((size_t)(R * C * K));
    const char *base = (const char *)a.data;
    for (int64_t r = 0; r < R; ++r)
        for (int64_t c = 0; c < C; ++c)
            for (int64_t k = 0; k < K; ++k)
                out[(size_t)((r * C + c) * K + k)] = *(const T *)(base + r * a.stride[0] + c * a.stride[1] + k * a.stride[2]);
}

template <typename T> static void scatter2(const std::vector<T> &in, const praline_array &a)
{
    const int64_t R = a.dim[0], C = a.dim[1];
    char *base = (char *)a.data;
    for (int64_t r = 0; r < R; ++r)
        for (int64_t c = 0; c < C; ++c) *(T *)(base + r * a.stride[0] + c * a.stride[1]) = in[(size_t)(r * C + c)];
}

template <typename T> static void scatter3(const std::vector<T> &in, const praline_array &a)
{
    const int64_t R = a.dim[0], C = a.dim[1], K = a.dim[2];
    char *base = (char *)a.data;
    for (int64_t r = 0; r < R; ++r)
        for (int64_t c = 0; c < C; ++c)
            for (int64_t k = 0; k < K; ++k)
                *(T *)(base + r * a.stride[0] + c * a.stride[1] + k * a.stride[2]) = in[(size_t)((r * C + c) * K + k)];
}

extern "C" int praline_build_scores(int num_sets, const praline_array *i1s, const praline_array *i2s,
                                    const praline_array *i1nzs, const praline_array *i2nzs, const praline_array *ss,
                                    const praline_array *m)
{
    (void)i1nzs; (void)i2nzs;  // dense contraction on the device; see praline_dp.h
    if (num_sets <= 0 || !i1s || !i2s || !ss || !arr_ok(m)) return fail(PRALINE_ERR_ARG, "NULL / empty build_scores argument");
    const int64_t L1 = i1s[0].dim[0], L2 = i2s[0].dim[0];
    if (L1 <= 0 || L2 <= 0) return fail(PRALINE_ERR_ARG, "empty sequence");
    if (m->dim[0] != L1 || m->dim[1] != L2) return fail(PRALINE_ERR_ARG, "m has shape %lldx%lld, expected %lldx%lld",
                                                     (long long)m->dim[0], (long long)m->dim[1], (long long)L1, (long long)L2);
    int64_t A = 0;
    for (int n = 0; n < num_sets; ++n) {
        if (!arr_ok(&i1s[n]) || !arr_ok(&i2s[n]) || !arr_ok(&ss[n])) return fail(PRALINE_ERR_ARG, "NULL array in set %d", n);
        if (i1s[n].dim[0] != L1 || i2s[n].dim[0] != L2) return fail(PRALINE_ERR_ARG, "set %d: profile lengths differ", n);
        if (ss[n].dim[0] != i1s[n].dim[1] || ss[n].dim[1] != i2s[n].dim[1])
            return fail(PRALINE_ERR_ARG, "set %d: score matrix shape does not match the profiles", n);
        A += std::max(i1s[n].dim[1], i2s[n].dim[1]);
    }
    if (A > 254) return fail(PRALINE_ERR_UNSUPPORTED, "concatenated alphabet size %lld > 254", (long long)A);
    // concatenate the track sets along the alphabet axis: P = [P_1 | P_2 ...], S = blockdiag(S_n)
    std::vector<float> prof((size_t)((L1 + L2) * A), 0.0f), S((size_t)(A * A), 0.0f), tmp;
    int64_t off = 0;
    for (int n = 0; n < num_sets; ++n) {
        const int64_t A1 = i1s[n].dim[1], A2 = i2s[n].dim[1];
        gather2<float>(i1s[n], tmp);
        for (int64_t r = 0; r < L1; ++r) for (int64_t c = 0; c < A1; ++c) prof[(size_t)(r * A + off + c)] = tmp[(size_t)(r * A1 + c)];
        gather2<float>(i2s[n], tmp);
        for (int64_t r = 0; r < L2; ++r) for (int64_t c = 0; c < A2; ++c) prof[(size_t)((L1 + r) * A + off + c)] = tmp[(size_t)(r * A2 + c)];
        gather2<float>(ss[n], tmp);
        for (int64_t r = 0; r < A1; ++r) for (int64_t c = 0; c < A2; ++c) S[(size_t)((off + r) * A + off + c)] = tmp[(size_t)(r * A2 + c)];
        off += std::max(A1, A2);
    }
    const int32_t lens[2] = {(int32_t)L1, (int32_t)L2};
    praline_arena *ar = nullptr;
    RC(praline_arena_create(2, lens, (int32_t)A, prof.data(), S.data(), &ar));
    DevBuf<float> d_m;
    int rc = d_m.alloc((size_t)(L1 * L2));
    if (rc == PRALINE_OK && (ar->wide || match_mode() == PRALINE_MATCH_REFERENCE)) {
        // the reference's own summation order (per track set), any alphabet: bit-identical to cext_build_scores
        std::vector<int32_t> sizes;
        for (int n = 0; n < num_sets; ++n) sizes.push_back((int32_t)std::max(i1s[n].dim[1], i2s[n].dim[1]));
        rc = praline_arena_set_track_sets(ar, num_sets, sizes.data());
        if (rc == PRALINE_OK) rc = arena_ensure_ref(ar);
        DevBuf<int32_t> d_pair, d_chunk;
        DevBuf<int64_t> d_off;
        if (rc == PRALINE_OK) rc = d_pair.upload(std::vector<int32_t>{0, 1}, g_rt.stream);
        if (rc == PRALINE_OK) rc = d_chunk.upload(std::vector<int32_t>{0}, g_rt.stream);
        if (rc == PRALINE_OK) rc = d_off.upload(std::vector<int64_t>{0}, g_rt.stream);
        if (rc == PRALINE_OK) rc = launch_match_ref(ar, d_pair.p, d_chunk.p, 1, (int)L1, d_off.p, d_m.p);
        if (rc == PRALINE_OK) {
            std::vector<float> hm((size_t)(L1 * L2));
            hipError_t e = hipMemcpyAsync(hm.data(), d_m.p, hm.size() * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(g_rt.stream);
            if (e != hipSuccess) rc = fail(PRALINE_ERR_DEVICE, "build_scores: %s", hipGetErrorString(e));
            else scatter2<float>(hm, *m);
        }
    } else if (rc == PRALINE_OK) {
        dim3 grid((unsigned)((L2 + 31) / 32), (unsigned)((L1 + 31) / 32));
        hipLaunchKernelGGL(k_scores_tile, grid, dim3(64), 0, g_rt.stream, ar->view(), 0, 1, ar->nstep, d_m.p);
        std::vector<float> hm((size_t)(L1 * L2));
        hipError_t e = hipMemcpyAsync(hm.data(), d_m.p, hm.size() * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g_rt.stream);
        if (e != hipSuccess) rc = fail(PRALINE_ERR_DEVICE, "build_scores: %s", hipGetErrorString(e));
        else scatter2<float>(hm, *m);
    }
    praline_arena_destroy(ar);
    return rc;
}

struct RawDev {
    DevBuf<float> m, g1, g2, o;
    DevBuf<uint8_t> t, z;
    int64_t L1 = 0, L2 = 0;
};

static int raw_upload(const praline_array *m, const praline_array *g1, const praline_array *g2, const praline_array *o,
                      const praline_array *t, const praline_array *z, RawDev &d)
{
    if (!arr_ok(m) || !arr_ok(g1) || !arr_ok(g2)) return fail(PRALINE_ERR_ARG, "NULL m / g1 / g2");
    const int64_t L1 = m->dim[0], L2 = m->dim[1];
    if (L1 <= 0 || L2 <= 0) return fail(PRALINE_ERR_ARG, "empty match score matrix");
    if (g1->dim[0] != L1 || g1->dim[1] != 2 || g2->dim[0] != L2 || g2->dim[1] != 2)
        return fail(PRALINE_ERR_ARG, "gap score arrays must be [L1][2] and [L2][2]");
    if (o && (o->dim[0] != L1 + 1 || o->dim[1] != L2 + 1 || o->dim[2] != 3)) return fail(PRALINE_ERR_ARG, "o must be [L1+1][L2+1][3]");
    if (t && (t->dim[0] != L1 + 1 || t->dim[1] != L2 + 1 || t->dim[2] != 3)) return fail(PRALINE_ERR_ARG, "t must be [L1+1][L2+1][3]");
    if (z && z->data && (z->dim[0] != L1 + 1 || z->dim[1] != L2 + 1)) return fail(PRALINE_ERR_ARG, "z must be [L1+1][L2+1]");
    RC(ensure_runtime(-1));
    d.L1 = L1; d.L2 = L2;
    hipStream_t st = g_rt.stream;
    std::vector<float> hm, hg1, hg2, ho;
    std::vector<uint8_t> ht, hz;
    gather2<float>(*m, hm); gather2<float>(*g1, hg1); gather2<float>(*g2, hg2);
    RC(d.m.upload(hm, st)); RC(d.g1.upload(hg1, st)); RC(d.g2.upload(hg2, st));
    const size_t cells = (size_t)((L1 + 1) * (L2 + 1));
    if (o) { gather3<float>(*o, ho); RC(d.o.upload(ho, st)); } else RC(d.o.alloc(cells * 3));
    if (t) { gather3<uint8_t>(*t, ht); RC(d.t.upload(ht, st)); } else RC(d.t.alloc(cells * 3));
    if (z && z->data) { gather2<uint8_t>(*z, hz); RC(d.z.upload(hz, st)); }
    else { RC(d.z.alloc(cells)); HIPCHK(hipMemsetAsync(d.z.p, 0, cells, st)); }
    HIPCHK(hipStreamSynchronize(st));
    return PRALINE_OK;
}

extern "C" int praline_align(int mode, const praline_array *m, const praline_array *g1, const praline_array *g2,
                             const praline_array *o, const praline_array *t, const praline_array *z)
{
    if (mode < 0 || mode > 4) return fail(PRALINE_ERR_ARG, "unknown alignment mode %d", mode);
    if (!arr_ok(o) || !arr_ok(t) || !arr_ok(z)) return fail(PRALINE_ERR_ARG, "NULL o / t / z");
    RawDev d;
    RC(raw_upload(m, g1, g2, o, t, z, d));
    hipLaunchKernelGGL(k_raw_align, dim3(1), dim3(64 * (unsigned)std::min<int64_t>(PRALINE_RAW_WAVES, std::max<int64_t>(1, (d.L2 + 63) / 64))), 0, g_rt.stream, mode == PRALINE_MODE_LOCAL ? 1 : 0, d.m.p, d.g1.p,
                       d.g2.p, d.o.p, d.t.p, d.z.p, (int)d.L1, (int)d.L2);
    HIPCHK(hipGetLastError());
    const size_t cells = (size_t)((d.L1 + 1) * (d.L2 + 1));
    std::vector<float> ho(cells * 3);
    std::vector<uint8_t> ht(cells * 3);
    HIPCHK(hipMemcpyAsync(ho.data(), d.o.p, ho.size() * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipMemcpyAsync(ht.data(), d.t.p, ht.size(), hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    scatter3<float>(ho, *o);
    scatter3<uint8_t>(ht, *t);
    return PRALINE_OK;
}

extern "C" int praline_align_global(const praline_array *m, const praline_array *g1, const praline_array *g2,
                                    const praline_array *o, const praline_array *t, const praline_array *z)
{ return praline_align(PRALINE_MODE_GLOBAL, m, g1, g2, o, t, z); }
extern "C" int praline_align_local(const praline_array *m, const praline_array *g1, const praline_array *g2,
                                   const praline_array *o, const praline_array *t, const praline_array *z)
{ return praline_align(PRALINE_MODE_LOCAL, m, g1, g2, o, t, z); }
extern "C" int praline_align_semiglobal_both(const praline_array *m, const praline_array *g1, const praline_array *g2,
                                             const praline_array *o, const praline_array *t, const praline_array *z)
{ return praline_align(PRALINE_MODE_SEMIGLOBAL_BOTH, m, g1, g2, o, t, z); }
extern "C" int praline_align_semiglobal_one(const praline_array *m, const praline_array *g1, const praline_array *g2,
                                            const praline_array *o, const praline_array *t, const praline_array *z)
{ return praline_align(PRALINE_MODE_SEMIGLOBAL_ONE, m, g1, g2, o, t, z); }
extern "C" int praline_align_semiglobal_two(const praline_array *m, const praline_array *g1, const praline_array *g2,
                                            const praline_array *o, const praline_array *t, const praline_array *z)
{ return praline_align(PRALINE_MODE_SEMIGLOBAL_TWO, m, g1, g2, o, t, z); }

extern "C" int praline_raw_align(int mode, const praline_array *m, const praline_array *g1, const praline_array *g2,
                                 const praline_array *z, float *score, int32_t *path, int64_t *path_rows)
{
    if (mode < 0 || mode > 4) return fail(PRALINE_ERR_ARG, "unknown alignment mode %d", mode);
    if (!score || !path || !path_rows) return fail(PRALINE_ERR_ARG, "NULL output");
    RawDev d;
    RC(raw_upload(m, g1, g2, nullptr, nullptr, z, d));
    hipStream_t st = g_rt.stream;
    const int L1 = (int)d.L1, L2 = (int)d.L2;
    hipLaunchKernelGGL(k_raw_init, dim3(256), dim3(256), 0, st, mode, d.g1.p, d.g2.p, d.o.p, d.t.p, L1, L2);
    hipLaunchKernelGGL(k_raw_align, dim3(1), dim3(64 * (unsigned)std::min<int64_t>(PRALINE_RAW_WAVES, std::max<int64_t>(1, ((int64_t)L2 + 63) / 64))), 0, st, mode == PRALINE_MODE_LOCAL ? 1 : 0, d.m.p, d.g1.p, d.g2.p,
                       d.o.p, d.t.p, d.z.p, L1, L2);
    DevBuf<float> d_score;
    DevBuf<int32_t> d_path;
    DevBuf<int64_t> d_info;
    const size_t cap = (size_t)(L1 + L2 + 2);
    RC(d_score.alloc(1)); RC(d_path.alloc(cap * 2)); RC(d_info.alloc(2));
    hipLaunchKernelGGL(k_raw_trace, dim3(1), dim3(256), 0, st, mode, d.o.p, d.t.p, L1, L2, d_score.p, d_path.p, d_info.p);
    HIPCHK(hipGetLastError());
    std::vector<int32_t> hp(cap * 2);
    int64_t info[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(score, d_score.p, sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hp.data(), d_path.p, hp.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(info, d_info.p, sizeof(info), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (info[1] <= 0 || info[0] < 0 || (size_t)(info[0] + info[1]) > cap) return fail(PRALINE_ERR_DEVICE, "traceback produced an invalid path");
    memcpy(path, hp.data() + 2 * info[0], (size_t)info[1] * 2 * sizeof(int32_t));
    *path_rows = info[1];
    return PRALINE_OK;
}

// --------------------------------------------------------------------------------------------
// debug: the per-lane match-score tile exactly as the fp32 MFMA chain forms it (NSTEP = arena.nstep via a
// runtime loop).  out: [64][32] floats.  Not part of the public header.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_debug_tile(ArenaDev ar, const int32_t *lane_one, int two0, int two1, int x0,
                                                    int y, int tp, int nstep, float *out)
{
    const int lane = threadIdx.x, half = lane >> 5, j = lane & 31;
    const int srcA = lane_one[j], srcB = lane_one[32 + j];
    const float *pA = ar.P + ((int64_t)(srcA >= 0 ? ar.row_off[srcA] : 0) + (y - 1)) * ar.KP + half * ar.KS;
    const float *pB = ar.P + ((int64_t)(srcB >= 0 ? ar.row_off[srcB] : 0) + (y - 1)) * ar.KP + half * ar.KS;
    const float *qA = ar.Q + ((int64_t)ar.row_off[two0] + x0 + j) * ar.KP + half * ar.KS;
    const float *qB = ar.Q + ((int64_t)ar.row_off[two1] + x0 + j) * ar.KP + half * ar.KS;
    f32x16 accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accB = accA;
    for (int k = 0; k < nstep; ++k) {
        accA = __builtin_amdgcn_mfma_f32_32x32x2f32(qA[k], pA[k], accA, 0, 0, 0);
        if (tp == 2) accB = __builtin_amdgcn_mfma_f32_32x32x2f32(qB[k], pB[k], accB, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float a = accA[r], b = (tp == 2) ? accB[r] : 0.0f;
        swap_halves(a, b);
        out[lane * 32 + 8 * (r >> 2) + (r & 3)] = a;
        out[lane * 32 + 8 * (r >> 2) + 4 + (r & 3)] = b;
    }
}

extern "C" int praline_debug_tile(praline_arena *arena, const int32_t *lane_one, int two0, int two1, int x0, int y, int tp,
                                  float *out)
{
    RC(arena_ready(arena));
    RC(ensure_runtime(-1));
    DevBuf<int32_t> d_l;
    DevBuf<float> d_o;
    std::vector<int32_t> lv(lane_one, lane_one + 64);
    RC(d_l.upload(lv, g_rt.stream));
    RC(d_o.alloc(64 * 32));
    hipLaunchKernelGGL(k_debug_tile, dim3(1), dim3(64), 0, g_rt.stream, arena->view(), d_l.p, two0, two1, x0, y, tp,
                       arena->nstep, d_o.p);
    HIPCHK(hipMemcpyAsync(out, d_o.p, 64 * 32 * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    return PRALINE_OK;
}

// --------------------------------------------------------------------------------------------
// diagnostics: the dense match-score matrix of one arena pair exactly as the kernels evaluate it
// --------------------------------------------------------------------------------------------
extern "C" int praline_arena_match_scores(praline_arena *arena, int32_t one, int32_t two, int kind, float *m)
{
    if (!arena || !m) return fail(PRALINE_ERR_ARG, "NULL argument");
    RC(arena_ready(arena));
    if (one < 0 || one >= arena->n_seqs || two < 0 || two >= arena->n_seqs) return fail(PRALINE_ERR_ARG, "index out of range");
    RC(ensure_runtime(-1));
    const int L1 = arena->len[one], L2 = arena->len[two];
    DevBuf<float> d_m;
    RC(d_m.alloc((size_t)L1 * L2));
    if (kind == 2) {   // the reference's summation order (what PRALINE_MATCH_REFERENCE plans and wide arenas use)
        RC(arena_ensure_ref(arena));
        DevBuf<int32_t> d_pair, d_chunk;
        DevBuf<int64_t> d_off;
        RC(d_pair.upload(std::vector<int32_t>{one, two}, g_rt.stream));
        RC(d_chunk.upload(std::vector<int32_t>{0}, g_rt.stream));
        RC(d_off.upload(std::vector<int64_t>{0}, g_rt.stream));
        RC(launch_match_ref(arena, d_pair.p, d_chunk.p, 1, L1, d_off.p, d_m.p));
        HIPCHK(hipMemcpyAsync(m, d_m.p, (size_t)L1 * L2 * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream));
        HIPCHK(hipStreamSynchronize(g_rt.stream));
        return PRALINE_OK;
    }
    if (arena->wide) return fail(PRALINE_ERR_UNSUPPORTED, "this arena has more than 32 active symbols: only kind 2 (reference order) exists");
    if (kind == 0) {
        dim3 grid((unsigned)((L2 + 31) / 32), (unsigned)((L1 + 31) / 32));
        hipLaunchKernelGGL(k_scores_tile, grid, dim3(64), 0, g_rt.stream, arena->view(), one, two, arena->nstep, d_m.p);
    } else if (kind == 1) {
        if (arena->nr16 == 0) return fail(PRALINE_ERR_UNSUPPORTED, "no f16 operands for this arena");
        int rc = praline_launch_scores_tile16(arena->view16(), arena->nr16, arena->nterm16, one, two, L1, L2, d_m.p, g_rt.stream);
        if (rc != PRALINE_OK) return fail(rc, "no k_scores_tile16 instance");
    } else return fail(PRALINE_ERR_ARG, "kind must be 0 (fp32 MFMA chain), 1 (f16 split) or 2 (reference order)");
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(m, d_m.p, (size_t)L1 * L2 * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    return PRALINE_OK;
}

extern "C" int praline_arena_info(const praline_arena *arena, int32_t *n_active, int32_t *mfma_steps_f32, int32_t *f16_ranges,
                                  int32_t *f16_terms)
{
    RC(arena_ready(arena));
    if (n_active) *n_active = arena->n_active;
    if (mfma_steps_f32) *mfma_steps_f32 = arena->nstep;
    if (f16_ranges) *f16_ranges = arena->nr16;
    if (f16_terms) *f16_terms = arena->nterm16;
    return PRALINE_OK;
}

extern "C" int praline_plan_tile_producer(const praline_plan *plan)
{
    if (!plan) return -1;
    if (plan->dense_kind == 1 && plan->arena->reft2_state != 1) return 2;   // (the arena no longer qualifies for k_match_tile)
    return plan->dense_kind;
}

// Which match-score arithmetic praline_plan_run uses for this plan: 0 = fp32 MFMA chain, 1 = f16 split.
extern "C" int praline_plan_match_kind(const praline_plan *plan)
{
    if (!plan) return -1;
    if (plan->dense_kind == 3) return 0;   // (per-position gap plans: the fp32 MFMA chain for both kinds of run)
    if (plan->dense_kind != 0) return 2;
    if (plan->split && plan->arena->nr16 > 0) {
        if (plan->want_paths) return 1;  // k_dp_split16_tb
        if (match_mode() != PRALINE_MATCH_F32) return 1;
    }
    return 0;
}
