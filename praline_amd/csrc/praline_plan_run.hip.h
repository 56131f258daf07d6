// praline_plan_run.hip.h -- part of praline_dp.hip (one translation unit; included there, in this order): praline_plan_run: kernel selection and launch sequences - scores (pipeline, shared waves, chain), dense-tile plans,
// alignments with paths (pipeline two-pass, packed int16, quad, strip kernels in chain / task / two-pass form), traceback.
// --------------------------------------------------------------------------------------------
// the scores kernels of the split-strip layout: k_dp_split16 on the f16 hi/lo operands, or - PRALINE_MM=f32 - k_dp_split on the
// fp32 MFMA chain (one translation unit per MFMA step count, dp_split_instance.hip); see dp_launch.hip.h
// --------------------------------------------------------------------------------------------
static int launch_scores(int nstep, const LaunchArgs &la, bool local)
{
    if (la.a16 != nullptr) {
        const int rc = praline_launch_split16(la, *la.a16, la.nr16, la.nterm16, local);
        if (rc != PRALINE_OK) return fail(rc, "no k_dp_split16 instance for nr=%d nterm=%d", la.nr16, la.nterm16);
        return PRALINE_OK;
    }
    switch (nstep) {
        case 2: return praline_launch_split_2(la, local);
        case 8: return praline_launch_split_8(la, local);
        case 10: return praline_launch_split_10(la, local);
        case 12: return praline_launch_split_12(la, local);
        case 14: return praline_launch_split_14(la, local);
        case 16: return praline_launch_split_16(la, local);
    }
    return fail(PRALINE_ERR_UNSUPPORTED, "no k_dp_split instance for nstep=%d", nstep);
}

// end cells of the semiglobal modes + device traceback for the tasks [t0, t1) of a path plan (after their fill)
static int launch_traceback(praline_plan &pl, const LaunchArgs &la, size_t t0, size_t t1, int mode)
{
    hipStream_t st = la.stream;
    // k_traceback runs over all pairs and skips those whose task is outside [t0, t1)
    const int threads = 64;
    const int64_t blocks = (pl.n_pairs + threads - 1) / threads;
    if (mode >= PRALINE_MODE_SEMIGLOBAL_BOTH) {   // end cells of the semiglobal modes, scanned per task
        const int64_t lanes = (int64_t)(t1 - t0) * (pl.quad ? 16 : (pl.split ? 32 : 64));   // (k_dp_pk16_tb writes the strip kernels' end-cell scratch)
        hipLaunchKernelGGL(k_semiglobal_end, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, la.ar,
                           pl.d_tasks.p, pl.d_lane_one.p, pl.d_lane_pair.p, pl.d_pairs.p, la.aux,
                           pl.d_end_cells.p, la.scores, la.rp, (int32_t)t0, (int32_t)t1, pl.quad ? 2 : (pl.split ? 1 : 0));
    }
    hipLaunchKernelGGL(k_traceback, dim3((unsigned)blocks), dim3(threads), 0, st, la.ar, pl.d_tasks.p,
                       pl.d_loc.p, pl.d_pairs.p, (const uint4 *)la.tb, la.aux, la.rl, pl.d_end_cells.p,
                       la.scores, pl.d_slot_off.p, pl.d_paths.p, pl.d_path_start.p, pl.d_path_rows.p, pl.n_pairs,
                       la.rp, (int32_t)t0, (int32_t)t1, pl.run_pk16 ? 3 : (pl.quad ? 2 : (pl.split ? 1 : 0)));
    HIPCHK(hipGetLastError());
    return PRALINE_OK;
}

// dense match-score tiles per launch chunk (bytes)
static size_t reftile_budget_bytes()
{
    if (const char *env = getenv("PRALINE_REFTILE_BUDGET_MB")) return (size_t)atoll(env) << 20;
    return (size_t)32 << 30;
}

// Plans whose DP reads its match scores from dense tiles (praline_plan::dense_kind: the reference's summation order, arenas
// without packed operands, many-rectangle plans on float profiles, per-position gap scores).  Per chunk of tasks one of the
// producers writes the tiles (4 bytes per cell and padding):
//   1  k_match_tile (dp_reftile.hip.h);  2  k_match_reft / k_match_ref, one cell per thread;  3  k_scores_tile_batch (fp32 MFMA)
// and a dense-tile DP instance consumes them: k_dp_split16<1, 1, LOCAL, 4> for scores, k_dp_split16_tb<1, 3, LOCAL, MASK, .., 4,
// PPG, NOFLAGS> (+ k_traceback) for alignments with paths, with per-position gap scores (ppg) and for the scores of tasks that
// are swept in several launches.  A task whose tile exceeds the chunk budget (sequences beyond ~16 000 positions) runs alone,
// a range of strips per launch: the tile then holds that range, the boundary column and the local maximum carry over.
// One stream, one tile set: a k_match_tile workgroup fills its CU (registers and LDS), so a second stream only time-slices the
// chip (measured on C2: four chunks alternating between two streams 31 ms, one chunk 19 ms).
static int plan_run_dense(praline_plan &pl, LaunchArgs la, Arena16Dev a16, int mode, bool local)
{
    praline_arena &a = *pl.arena;
    int producer = pl.dense_kind;
    if (producer == 1) {
        if (a.reft2_state == 0) RC(arena_ensure_reft2(&a));   // (the arena changed since the plan was made)
        if (a.reft2_state != 1) producer = 2;                  // (... and no longer qualifies for k_match_tile)
    }
    hipStream_t st = g_rt.stream;
    const size_t nt = pl.tasks.size();
    const bool semiglobal = mode >= 2;
    const bool ppg = pl.run_ppg;
    const size_t m_budget = reftile_budget_bytes(), tb_budget = tb_budget_bytes();
    auto strip_floats = [&](const WaveTask &wt) { return (int64_t)(wt.max_l1 + PRALINE_DENSE_PAD) * 1024; };
    // range: the chunk is ONE task and sweeps its strips [strip_lo, strip_lo + strip_cnt); last: the task's end cells are final
    struct Chunk { size_t t0, t1, b0, b1, c0, c1; int64_t m_e, tb_e, aux_e; int strip_lo, strip_cnt, max_l1, strips; bool range, last; };
    std::vector<Chunk> chunks;
    std::vector<int64_t> dense_off(nt, 0);
    std::vector<RefTileBlock> blocks;
    std::vector<int32_t> grp;   // group records (dp_reftile.h)
    std::vector<int32_t> chunk_pairs;   // the pairs of every chunk, chunk after chunk (producers 2 and 3)
    bool any_range = false;
    for (size_t t = 0; t < nt; ++t) any_range = any_range || (size_t)(pl.tasks[t].nstrips * strip_floats(pl.tasks[t])) * 4 > m_budget;
    // plans without paths: the scores kernel, unless a task runs in strip ranges or with per-position gap scores
    const bool fill_only = !pl.want_paths && (ppg || any_range);
    // k_match_tile's workgroups of a chunk: the tasks are grouped by their 32 sequences one (the schedule gives every
    // sequence two of a set of ones its own task), a group's sequences two are laid end to end and cut into 128 columns
    auto add_blocks = [&](size_t t0, size_t t1) {
        std::unordered_map<std::string, size_t> index;
        std::vector<std::vector<int32_t>> members;
        for (size_t t = t0; t < t1; ++t) {
            const WaveTask &wt = pl.tasks[t];
            if (wt.max_l1 <= 0 || wt.two[0] < 0 || a.len[(size_t)wt.two[0]] <= 0) continue;
            const std::string key(reinterpret_cast<const char *>(pl.h_lane_one.data() + t * 32), 32 * sizeof(int32_t));
            auto it = index.find(key);
            if (it == index.end()) { it = index.emplace(key, members.size()).first; members.emplace_back(); }
            members[it->second].push_back((int32_t)(t - t0));
        }
        for (const std::vector<int32_t> &mem : members) {
            const int32_t base = (int32_t)grp.size();
            int32_t cum = 0;
            // (whole strips: the columns between the end of a sequence and the end of its last strip receive zeros - local
            // alignments must not see stale positive scores there)
            for (int32_t tr : mem) { grp.push_back(cum); cum += (a.len[(size_t)pl.tasks[t0 + (size_t)tr].two[0]] + 31) / 32 * 16; }
            grp.push_back(cum);
            grp.insert(grp.end(), mem.begin(), mem.end());
            for (int32_t c = 0; c * 64 < cum; ++c) blocks.push_back({base, (int32_t)mem.size(), c, 0});
        }
    };
    auto add_pairs = [&](size_t t0, size_t t1, int &max_l1, int &strips) {
        for (size_t t = t0; t < t1; ++t) {
            bool any = false;
            for (int l = 0; l < 32; ++l) {
                const int32_t p = pl.h_lane_pair[t * 32 + l];
                if (p < 0) continue;
                chunk_pairs.push_back(p);
                any = true;
            }
            if (any) { max_l1 = std::max(max_l1, (int)pl.tasks[t].max_l1); strips = std::max(strips, (int)pl.tasks[t].nstrips); }
        }
    };
    int64_t bnd4_e = 0;   // fill_only: the float4 boundary columns of k_dp_split16_tb (the plan's own are float2)
    for (size_t t0 = 0; t0 < nt;) {
        const WaveTask &w0 = pl.tasks[t0];
        const int64_t sf = strip_floats(w0);
        if ((size_t)(w0.nstrips * sf) * 4 > m_budget) {
            // one task, strip ranges
            const int per = (int)std::max<int64_t>(1, (int64_t)(m_budget / 4) / sf);
            const size_t c0 = chunk_pairs.size();
            int ml = 0, strips = 0;
            add_pairs(t0, t0 + 1, ml, strips);
            dense_off[t0] = 0;
            pl.tasks[t0].tb_off = 0;
            pl.tasks[t0].aux_off = 0;
            if (fill_only) { pl.tasks[t0].bnd_off = bnd4_e; bnd4_e += (int64_t)(w0.max_l1 + 24) * 32; }
            for (int lo = 0; lo < w0.nstrips; lo += per) {
                const int cnt = std::min(per, w0.nstrips - lo);
                chunks.push_back({t0, t0 + 1, blocks.size(), blocks.size(), c0, chunk_pairs.size(), (int64_t)cnt * sf,
                                  pl.want_paths ? pl.tb_elems[t0] : 0, semiglobal ? pl.aux_elems[t0] : 0, lo, cnt, ml, cnt, true,
                                  lo + cnt >= w0.nstrips});
            }
            ++t0;
            continue;
        }
        size_t t1 = t0;
        const size_t b0 = blocks.size(), c0 = chunk_pairs.size();
        int64_t m_e = 0, tb_e = 0, aux_e = 0;
        while (t1 < nt) {
            const WaveTask &wt = pl.tasks[t1];
            const int64_t m_add = wt.nstrips * strip_floats(wt), tb_add = pl.want_paths ? pl.tb_elems[t1] : 0;
            if ((size_t)m_add * 4 > m_budget) break;   // (the next task runs alone)
            if (t1 > t0 && ((size_t)(m_e + m_add) * 4 > m_budget || (size_t)(tb_e + tb_add) * 8 > tb_budget)) break;
            dense_off[t1] = m_e;
            pl.tasks[t1].tb_off = tb_e;
            pl.tasks[t1].aux_off = aux_e;
            if (fill_only) { pl.tasks[t1].bnd_off = bnd4_e; bnd4_e += (int64_t)(wt.max_l1 + 24) * 32; }
            m_e += m_add;
            tb_e += tb_add;
            aux_e += ((pl.want_paths || fill_only) && semiglobal) ? pl.aux_elems[t1] : 0;
            ++t1;
        }
        int ml = 0, strips = 0;
        if (producer == 1) add_blocks(t0, t1);
        else add_pairs(t0, t1, ml, strips);
        chunks.push_back({t0, t1, b0, blocks.size(), c0, chunk_pairs.size(), m_e, tb_e, aux_e, 0, 0x3fffffff, ml, strips, false, true});
        t0 = t1;
    }
    if (producer != 3 && (producer == 2 || any_range)) { RC(arena_ensure_ref(&a)); }
    {
        size_t need_m = 1, need_tb = 0, need_ax = 1;
        for (const Chunk &ch : chunks) {
            need_m = std::max(need_m, (size_t)ch.m_e);
            need_tb = std::max(need_tb, (size_t)ch.tb_e * 8);
            need_ax = std::max(need_ax, (size_t)ch.aux_e);
        }
        // (every buffer is sized once, before the loop: see the chunk loops of praline_plan_run)
        if (pl.d_dense.n < need_m) RC(pl.d_dense.alloc(need_m));
        if (pl.want_paths && pl.d_tb.n < need_tb) RC(pl.d_tb.alloc(need_tb));
        if ((pl.want_paths || fill_only) && pl.d_aux.n < need_ax) RC(pl.d_aux.alloc(need_ax));
    }
    if (fill_only) {
        if (pl.d_bnd_chain.n < (size_t)bnd4_e * sizeof(float4)) RC(pl.d_bnd_chain.alloc((size_t)bnd4_e * sizeof(float4)));
        if (pl.d_end_cells.n < (size_t)pl.n_pairs * 4) RC(pl.d_end_cells.alloc((size_t)pl.n_pairs * 4));
    }
    if (!pl.d_tasks.p) RC(pl.d_tasks.alloc(nt));
    if (pl.d_dense_off.n < nt) RC(pl.d_dense_off.alloc(nt));
    if (pl.d_tile_blocks.n < blocks.size()) RC(pl.d_tile_blocks.alloc(std::max<size_t>(blocks.size(), 1)));
    if (pl.d_tile_grp.n < grp.size()) RC(pl.d_tile_grp.alloc(std::max<size_t>(grp.size(), 1)));
    if (pl.d_chunk_pairs.n < chunk_pairs.size()) RC(pl.d_chunk_pairs.alloc(std::max<size_t>(chunk_pairs.size(), 1)));
    HIPCHK(hipMemcpyAsync(pl.d_tasks.p, pl.tasks.data(), nt * sizeof(WaveTask), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(pl.d_dense_off.p, dense_off.data(), nt * sizeof(int64_t), hipMemcpyHostToDevice, st));
    if (!grp.empty()) HIPCHK(hipMemcpyAsync(pl.d_tile_grp.p, grp.data(), grp.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (!blocks.empty())
        HIPCHK(hipMemcpyAsync(pl.d_tile_blocks.p, blocks.data(), blocks.size() * sizeof(RefTileBlock), hipMemcpyHostToDevice, st));
    if (!chunk_pairs.empty())
        HIPCHK(hipMemcpyAsync(pl.d_chunk_pairs.p, chunk_pairs.data(), chunk_pairs.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));   // (the host lists go out of scope)
    {
        char kn[200];
        const char *lb = local ? "true" : "false";
        if (pl.want_paths || fill_only)
            snprintf(kn, sizeof(kn), "k_dp_split16_tb<1, 3, %s, %s, false, false, 4, %s, %s>", lb, pl.has_rects ? "true" : "false",
                     ppg ? "true" : "false", fill_only ? "true" : "false");
        else snprintf(kn, sizeof(kn), "k_dp_split16<1, 1, %s, 4, 1, false>", lb);   // (four-wave workgroups: refined below)
        pl.last_kernel = kn;
    }
    for (const Chunk &ch : chunks) {
        // ---- the tiles ----
        const int chunk_producer = (producer == 1 && ch.range) ? 2 : producer;
        if (chunk_producer == 1) {
            RefTileArgs g;
            g.raw = a.d_raw.p;
            g.A = a.A;
            g.T2 = a.d_reft2.p;
            g.PR = a.pair_rows;
            g.row_off_raw = a.d_row_off_raw.p;
            g.len = a.d_len.p;
            g.pr_off = a.d_pr_off.p;
            g.set_lo = a.d_set_lo.p;
            g.n_sets = (int)a.set_lo.size() - 1;
            g.tasks = pl.d_tasks.p + ch.t0;
            g.lane_one = pl.d_lane_one.p + ch.t0 * 32;
            g.dense_off = pl.d_dense_off.p + ch.t0;
            g.m = pl.d_dense.p;
            g.blocks = pl.d_tile_blocks.p + ch.b0;
            g.grp = pl.d_tile_grp.p;
            g.waves = 0;
            int rc = praline_launch_match_tile(g, a.ref_tb, (unsigned)(ch.b1 - ch.b0), st);
            if (rc != PRALINE_OK) return fail(rc, "k_match_tile launch failed (A=%d, tb=%d)", a.A, a.ref_tb);
        } else if (ch.c1 > ch.c0) {
            TileOut to;
            to.loc = pl.d_loc.p;
            to.tasks = pl.d_tasks.p;
            to.dense_off = pl.d_dense_off.p;
            to.strip_lo = ch.strip_lo;
            to.strip_cnt = ch.strip_cnt;
            if (chunk_producer == 2) {
                RC(launch_match_ref(&a, pl.d_pairs.p, pl.d_chunk_pairs.p + ch.c0, ch.c1 - ch.c0, ch.max_l1, nullptr, pl.d_dense.p, to));
            } else {
                const int tiles_x = ch.strips, tiles_y = (ch.max_l1 + 31) / 32;
                if (tiles_x > 0 && tiles_y > 0) {
                    hipLaunchKernelGGL(k_scores_tile_batch, dim3((unsigned)(ch.c1 - ch.c0), (unsigned)tiles_y), dim3(64), 0, st,
                                       a.view(), pl.d_pairs.p, pl.d_chunk_pairs.p + ch.c0, nullptr, a.nstep, tiles_x, pl.d_dense.p, to);
                    HIPCHK(hipGetLastError());
                }
            }
        }
        // ---- the fill ----
        a16.dense = pl.d_dense.p;
        a16.dense_off = pl.d_dense_off.p + ch.t0;
        la.stream = st;
        la.tasks = pl.d_tasks.p + ch.t0;
        la.lane_one = pl.d_lane_one.p + ch.t0 * 32;
        la.lane_pair = pl.d_lane_pair.p + ch.t0 * 32;
        la.n_tasks = (unsigned)(ch.t1 - ch.t0);
        la.bnd = fill_only ? (void *)pl.d_bnd_chain.p : (void *)pl.d_bnd.p;
        int rc;
        if (!pl.want_paths && !fill_only) {
            la.tb = nullptr;
            la.aux = nullptr;
            la.wg = nullptr;
            la.n_wg = 0;
            if (chunks.size() == 1 && !pl.wg.empty() && !(getenv("PRALINE_NO_W2") && getenv("PRALINE_NO_W2")[0] == '1')) {
                // (the shared-wave descriptors index the plan's task list: one chunk only)
                if (!pl.d_wg.p) { RC(pl.d_wg.upload(pl.wg, st)); }
                la.wg = pl.d_wg.p;
                la.n_wg = (unsigned)pl.wg.size();
                char kn[160];
                snprintf(kn, sizeof(kn), "k_dp_split16<1, 1, %s, 4, 4, false>", local ? "true" : "false");
                pl.last_kernel = kn;
            }
            rc = praline_launch_dense(la, a16, local);
            if (rc != PRALINE_OK) return fail(rc, "dense-tile scores launch failed");
            continue;
        }
        la.tb = (uint4 *)pl.d_tb.p;
        la.aux = pl.d_aux.p;
        la.end_cells = pl.d_end_cells.p;
        rc = praline_launch_dense_tb(la, a16, local, pl.has_rects, ppg, fill_only, ch.strip_lo, ch.strip_cnt);
        if (rc != PRALINE_OK) return fail(rc, "dense-tile fill launch failed");
        if (!ch.last) continue;
        if (pl.want_paths) {
            RC(launch_traceback(pl, la, ch.t0, ch.t1, mode));
        } else if (semiglobal) {
            const int64_t lanes = (int64_t)(ch.t1 - ch.t0) * 32;
            hipLaunchKernelGGL(k_semiglobal_end, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, la.ar, pl.d_tasks.p, pl.d_lane_one.p,
                               pl.d_lane_pair.p, pl.d_pairs.p, la.aux, pl.d_end_cells.p, la.scores, la.rp, (int32_t)ch.t0, (int32_t)ch.t1, 1);
            HIPCHK(hipGetLastError());
        }
    }
    return PRALINE_OK;
}

// Score plans of a FEW LONG sequences: the score kernels put at most four waves on a task (its strips form a chain), so a
// plan of a handful of tasks leaves the chip idle - a single 30 000 x 30 000 alignment took 2.7 s scores-only and 54 ms
// with paths.  Such plans run the chain-mode fill (one wave per task and strip, pipelined across workgroups) in its
// flag-free form (k_dp_split16_tb<..., CHAIN, TWOPASS>): same scores bit for bit.  The choice is an estimate from the
// schedule, fitted to scripts/exp_scores_chain.py (all pairs of N x ~mu residues; chain wins from single alignments up to
// about N = 64 x 400 and for every batch of long sequences that the pipeline workgroups do not take):
//   shared waves: the longest task's ceil(strips / 4) x rows steps at 0.55 us (one-hot lookup instances: 0.45), in rounds
//                 of 2048 waves;
//   chain:        the larger of the longest task's rows + 24 x strips steps at 0.7 us and an even share of all strip-rows
//                 over 2048 waves at 2.0 us per step (the waves of a chain wait for each other).
static bool plan_scores_chain_wanted(const praline_plan &pl)
{
    if (const char *env = getenv("PRALINE_SCORES_CHAIN")) return atoi(env) != 0;
    const size_t nt = pl.tasks.size();
    if (nt == 0 || (int64_t)nt > chain_max_tasks()) return false;
    double shared = 0.0, crit = 0.0, work = 0.0, bnd_bytes = 0.0;
    int max_strips = 0;
    for (const WaveTask &wt : pl.tasks) {
        const double rows = wt.max_l1 + 1.0;
        shared = std::max(shared, std::ceil(wt.nstrips / 4.0) * rows);
        crit = std::max(crit, rows + 24.0 * wt.nstrips);
        work += wt.nstrips * rows;
        bnd_bytes += (wt.nstrips + 1.0) * (wt.max_l1 + 24.0) * 512.0;
        max_strips = std::max(max_strips, (int)wt.nstrips);
    }
    if (max_strips < 2 || bnd_bytes > 64.0 * 1073741824.0) return false;
    const bool lookup = pl.arena->onehot && pl.arena->nterm16 == 1;
    const double t_shared = shared * std::ceil(4.0 * nt / 2048.0) * (lookup ? 0.45 : 0.55);   // us
    const double t_chain = std::max(crit * 0.7, work / 2048.0 * 2.0);
    return t_chain < 0.9 * t_shared;
}

static int plan_run_scores_chain(praline_plan &pl, LaunchArgs la, const Arena16Dev &a16, int mode, bool local)
{
    praline_arena &a = *pl.arena;
    hipStream_t st = g_rt.stream;
    const size_t nt = pl.tasks.size();
    const bool semiglobal = mode >= 2;
    std::vector<WaveTask> ct(pl.tasks.begin(), pl.tasks.end());
    int64_t bnd_e = 0, aux_e = 0;
    int max_strips = 0, rows = 0;
    for (size_t t = 0; t < nt; ++t) {
        WaveTask &wt = ct[t];
        wt.bnd_off = bnd_e;
        wt.tb_off = 0;
        wt.aux_off = aux_e;
        bnd_e += (int64_t)(wt.nstrips + 1) * (wt.max_l1 + 24) * 32;   // float4 elements, [strip boundary][row][32]
        aux_e += semiglobal ? pl.aux_elems[t] : 0;
        max_strips = std::max(max_strips, (int)wt.nstrips);
        rows = std::max(rows, (int)wt.max_l1);
    }
    const size_t n_flags = nt * (size_t)(max_strips + 1);
    if (pl.d_bnd_chain.n < (size_t)bnd_e * sizeof(float4)) RC(pl.d_bnd_chain.alloc((size_t)bnd_e * sizeof(float4)));
    if (pl.d_chain_flags.n < n_flags) RC(pl.d_chain_flags.alloc(n_flags));
    if (local && pl.d_chain_cand.n < n_flags * 32) RC(pl.d_chain_cand.alloc(n_flags * 32));
    if (pl.d_aux.n < (size_t)std::max<int64_t>(aux_e, 1)) RC(pl.d_aux.alloc((size_t)std::max<int64_t>(aux_e, 1)));
    if (pl.d_end_cells.n < (size_t)pl.n_pairs * 4) RC(pl.d_end_cells.alloc((size_t)pl.n_pairs * 4));
    if (pl.d_tasks_chain.n < nt) RC(pl.d_tasks_chain.alloc(nt));
    if (semiglobal && !pl.d_pairs.p) {
        if (pl.h_pairs.size() != (size_t)pl.n_pairs * 2) return fail(PRALINE_ERR_UNSUPPORTED, "score plan without its pair list");
        RC(pl.d_pairs.upload(pl.h_pairs, st));
    }
    HIPCHK(hipMemsetAsync(pl.d_chain_flags.p, 0, n_flags * sizeof(int), st));
    HIPCHK(hipMemcpyAsync(pl.d_tasks_chain.p, ct.data(), nt * sizeof(WaveTask), hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));   // ct goes out of scope
    la.tasks = pl.d_tasks_chain.p;
    la.n_tasks = (unsigned)nt;
    la.bnd = pl.d_bnd_chain.p;
    la.tb = nullptr;
    la.aux = pl.d_aux.p;
    la.end_cells = pl.d_end_cells.p;
    la.stream = st;
    int every = nt >= 512 ? 96 : (nt >= 64 ? 24 : 6);
    every = std::min(every, std::max(6, rows / 4));
    if (const char *env = getenv("PRALINE_CHAIN_EVERY")) every = std::max(6, atoi(env));
    int rc = praline_launch_scores_chain(la, a16, a.nr16, a.nterm16, local, max_strips, pl.d_chain_flags.p, pl.d_chain_cand.p, every);
    if (rc != PRALINE_OK) return fail(rc, "no scores-only chain instance for nr=%d nterm=%d", a.nr16, a.nterm16);
    if (local) {
        const int64_t lanes = (int64_t)nt * 32;
        hipLaunchKernelGGL(k_chain_local_end, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, la.tasks, la.lane_pair,
                           pl.d_chain_cand.p, (int)nt, max_strips + 1, pl.d_end_cells.p, la.scores);
    }
    if (semiglobal) {
        const int64_t lanes = (int64_t)nt * 32;
        hipLaunchKernelGGL(k_semiglobal_end, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, la.ar, la.tasks, la.lane_one,
                           la.lane_pair, pl.d_pairs.p, la.aux, pl.d_end_cells.p, la.scores, la.rp, (int32_t)0, (int32_t)nt, 1);
    }
    HIPCHK(hipGetLastError());
    return PRALINE_OK;
}

extern "C" int praline_plan_run(praline_plan *plan, int mode, float gap_open, float gap_extend, void *d_scores)
{
    if (!plan) return fail(PRALINE_ERR_ARG, "plan is NULL");
    if (mode < 0 || mode > 4) return fail(PRALINE_ERR_ARG, "unknown alignment mode %d", mode);
    if (!(gap_open <= 0.0f) || !(gap_extend <= 0.0f))
        return fail(PRALINE_ERR_UNSUPPORTED, "batched kernels need gap scores <= 0 (got %g, %g)", gap_open, gap_extend);
    RC(ensure_runtime(-1));
    praline_plan &pl = *plan;
    if (pl.n_pairs == 0) return PRALINE_OK;
    const praline_arena &a = *pl.arena;
    LaunchArgs la;
    la.wg = nullptr;
    la.n_wg = 0;
    la.ar = a.view();
    la.lane_one = pl.d_lane_one.p;
    la.lane_pair = pl.d_lane_pair.p;
    la.bnd = pl.d_bnd.p;
    la.rl.rect_off = pl.has_rects ? pl.d_rect_off.p : nullptr;
    la.rl.rects = pl.has_rects ? pl.d_rects.p : nullptr;
    la.rl.zmask = pl.mask_kind == 2 ? pl.d_zmask.p : nullptr;
    la.rl.zm_off = pl.mask_kind == 2 ? pl.d_zm_off.p : nullptr;
    la.scores = d_scores ? (float *)d_scores : pl.d_scores.p;
    la.end_cells = pl.d_end_cells.p;
    la.rp.mode = mode;
    la.rp.go1 = la.rp.go2 = gap_open;
    la.rp.ge1 = la.rp.ge2 = gap_extend;
    la.stream = g_rt.stream;
    la.split = pl.split ? 1 : 0;
    // match scores on the matrix pipe (f16 hi/lo split) unless PRALINE_MM=f32 asks for the fp32 MFMA chain
    Arena16Dev a16 = a.view16();
    la.a16 = nullptr;
    la.nr16 = a.nr16;
    la.nterm16 = a.nterm16;
    if (pl.split && a.nr16 > 0 && match_mode() != PRALINE_MATCH_F32) la.a16 = &a16;
    const bool local = mode == PRALINE_MODE_LOCAL;
    pl.last_mode = mode;
    pl.last_scores = la.scores;
    hipStream_t st = g_rt.stream;

    {
        char kn[160];
        const char *lb = local ? "true" : "false";
        if (pl.want_paths && pl.quad) snprintf(kn, sizeof(kn), "k_dp_quad_tb<%d, ...>", a.nr16);
        else if (pl.want_paths && pl.pk16) snprintf(kn, sizeof(kn), "k_dp_pk16_tb<%d, ...>", a.nr16);   // (refined below: a run may take the strip kernels)
        else if (pl.want_paths) snprintf(kn, sizeof(kn), "k_dp_split16_tb<%d, ...>", a.nr16);   // refined below (nterm, chain)
        else if (la.a16 == nullptr) snprintf(kn, sizeof(kn), "k_dp_split<%d, %s>", a.nstep, lb);
        else {
            const char *nl = getenv("PRALINE_NO_LOOKUP");
            const bool lookup = !(nl && nl[0] == '1');
            const bool shared = !pl.wg.empty() && a16.stage;
            const bool table = a.nterm16 == 1 && a16.sym8 != nullptr && (!shared || lookup);   // one-hot path (lookup or operand table)
            const bool four = (shared && (!table || lookup)) || (a16.stage && !table && !pl.wg_singles.empty() &&
                              !(getenv("PRALINE_NO_W2") && getenv("PRALINE_NO_W2")[0] == '1'));
            snprintf(kn, sizeof(kn), "k_dp_split16<%d, %d, %s, %d, %d, false>", a.nr16, a.nterm16, lb,
                     table ? (lookup ? 3 : 1) : (a16.stage ? 2 : 0), four ? 4 : 1);
        }
        pl.last_kernel = kn;
    }
    if (pl.run_ppg) la.rp.gaps = a.d_gaps.p;
    if (pl.dense_kind != 0) {   // (names the kernel it launches)
        HIPCHK(hipEventRecord(pl.ev0, st));
        RC(plan_run_dense(pl, la, a16, mode, local));
        HIPCHK(hipEventRecord(pl.ev1, st));
        return PRALINE_OK;
    }
    if (!pl.want_paths && pl.pipe.ok) {   // (the match-score mode was read when the plan was created)
        char kn[160];
        snprintf(kn, sizeof(kn), "k_dp_pipe<%d, %d, %s, %s, false>", a.nr16, a.nterm16, local ? "true" : "false", mode >= 2 ? "true" : "false");
        pl.last_kernel = kn;
        PipeLaunch pp;
        pp.items = pl.d_pipe_items.p;
        pp.n_items = (unsigned)pl.pipe.items.size();
        pp.tasks = pl.d_pipe_tasks.p;
        pp.set_one = pl.d_pipe_set_one.p;
        pp.lane_pair = pl.d_pipe_lane_pair.p;
        pp.bnd = pl.d_pipe_bnd.p;
        pp.analytic = pl.d_pipe_analytic.p;
        pp.analytic_rows = pl.pipe_analytic_rows;
        pp.analytic_valid = pl.pipe_analytic_mode == mode && pl.pipe_analytic_go == la.rp.go1 && pl.pipe_analytic_ge == la.rp.ge1;
        pl.pipe_analytic_mode = mode; pl.pipe_analytic_go = la.rp.go1; pl.pipe_analytic_ge = la.rp.ge1;
        pp.scores = la.scores;
        pp.rp = la.rp;
        pp.stream = st;
        HIPCHK(hipEventRecord(pl.ev0, st));
        RC(praline_launch_pipe(pp, a16, a.nr16, a.nterm16));
        HIPCHK(hipEventRecord(pl.ev1, st));
        HIPCHK(hipGetLastError());
        return PRALINE_OK;
    }
    if (!pl.want_paths && pl.split && la.a16 != nullptr && !pl.has_rects) {
        if (pl.scores_chain < 0) pl.scores_chain = plan_scores_chain_wanted(pl) ? 1 : 0;
        if (pl.scores_chain == 1) {
            char kn[160];
            snprintf(kn, sizeof(kn), "k_dp_split16_tb<%d, %d, %s, false, true, true, 0>", a.nr16, a.nterm16, local ? "true" : "false");
            pl.last_kernel = kn;
            HIPCHK(hipEventRecord(pl.ev0, st));
            RC(plan_run_scores_chain(pl, la, a16, mode, local));
            HIPCHK(hipEventRecord(pl.ev1, st));
            return PRALINE_OK;
        }
    }
    if (!pl.want_paths) {
        if (!pl.d_tasks.p) { RC(pl.d_tasks.upload(pl.tasks, st)); }
        la.tasks = pl.d_tasks.p;
        la.tb = nullptr;
        la.aux = nullptr;
        la.n_tasks = (unsigned)pl.tasks.size();
        if (!pl.wg.empty() && la.a16 != nullptr && a16.stage) {
            // small batch: shared-wave workgroups on the staged stream - also for one-hot arenas (measured,
            // 1024 tasks: 3520 vs 3099 GCUPS; the one-hot table path wins, by 4 %, only on a full chip)
            if (!pl.d_wg.p) { RC(pl.d_wg.upload(pl.wg, st)); }
            la.wg = pl.d_wg.p;
            la.n_wg = (unsigned)pl.wg.size();
            // one-hot arenas keep their symbol stream: the shared waves look their match scores up (BSRC = 3);
            // PRALINE_NO_LOOKUP=1: the staged operand stream as for float profiles
            const char *nl = getenv("PRALINE_NO_LOOKUP");
            if (a.nterm16 != 1 || (nl && nl[0] == '1')) a16.sym8 = nullptr;
        }
        // large batches: four independent tasks per workgroup (wg_singles) for arenas without the one-hot table
        // (measured, float profiles: +0..6 %); one-hot arenas are faster on the table path in every mode (C4 rank
        // share: 4.4 TCUPS global, 4.0 local against 3.1 on this list)
        else if (a16.sym8 == nullptr && !pl.wg_singles.empty() && la.a16 != nullptr && a16.stage &&
                 !(getenv("PRALINE_NO_W2") && getenv("PRALINE_NO_W2")[0] == '1')) {
            if (!pl.d_wg_singles.p) { RC(pl.d_wg_singles.upload(pl.wg_singles, st)); }
            la.wg = pl.d_wg_singles.p;
            la.n_wg = (unsigned)pl.wg_singles.size();
        }
        HIPCHK(hipEventRecord(pl.ev0, st));
        RC(launch_scores(a.nstep, la, local));
        HIPCHK(hipEventRecord(pl.ev1, st));
        HIPCHK(hipGetLastError());
        return PRALINE_OK;
    }

    // ---- with paths: chunk the tasks so the packed traceback fits the scratch budget ----
    // k_dp_split16_tb's single-term instances take the tie flags from the predecessor states instead of the candidate
    // sums (dp_split16_tb.hip.h, INTS): valid when every DP value is a multiple of 2^-k that float32 holds exactly -
    // one-hot profiles, S and gap scores integral after scaling by 2^k, (L1 + L2) * max |score| * 2^k < 2^24.
    // Other exact-mode arenas run the three-term instances (their lo pieces are zero: same match scores).
    int tb_nterm = a.nterm16;
    pl.run_pk16 = false;
    float pk16_scale = 1.0f;
    if (a.nterm16 == 1) {
        double big_scaled = 1e30;
        int k_bits = 0;
        bool ints = a.all_onehot && a.s_scale_bits >= 0 && !(getenv("PRALINE_NO_INTS") && getenv("PRALINE_NO_INTS")[0] == '1');
        if (ints) {
            int k = a.s_scale_bits;
            for (; k <= 8; ++k) {
                const float sc = (float)(1 << k), g1 = gap_open * sc, g2 = gap_extend * sc;
                if (std::isfinite(g1) && std::isfinite(g2) && g1 == std::nearbyint(g1) && g2 == std::nearbyint(g2)) break;
            }
            const double big = std::max((double)a.s_absmax, std::max(std::fabs((double)gap_open), std::fabs((double)gap_extend)));
            ints = k <= 8 && (2.0 * a.max_len + 4.0) * big * (double)(1 << std::min(k, 8)) < 16777216.0;
            k_bits = k;
            big_scaled = big * (double)(1 << std::min(k, 8));
        }
        tb_nterm = ints ? 1 : 3;
        // two pairs per lane in int16 when every DP value of this run fits (dp_pk16.hip.h)
        pl.run_pk16 = pl.want_paths && pl.pk16 && ints && (2.0 * a.max_len + 36.0) * big_scaled < 32000.0;
        pk16_scale = (float)(1 << std::min(std::max(k_bits, 0), 8));
    }
    // rectangle slots per pair the packed kernel holds in registers: the lists' longest, or the slots filled so far
    // (praline_plan_mask_path_bounds), rounded up to an instance (1, 2, PRALINE_MAX_RECTS)
    int pk16_slots = !pl.has_rects ? 0 : (pl.slot_rects >= 0 ? pl.slot_rects : pl.max_rects);
    pk16_slots = pk16_slots <= 0 ? (pl.has_rects ? 1 : 0) : (pk16_slots <= 2 ? pk16_slots : PRALINE_MAX_RECTS);
    if (pl.want_paths && pl.pk16) {
        char kn[160];
        if (pl.run_pk16) snprintf(kn, sizeof(kn), "k_dp_pk16_tb<%d, %s, %d, false>", a.nr16, local ? "true" : "false", pk16_slots);   // (chain mode: below)
        else snprintf(kn, sizeof(kn), "k_dp_split16_tb<%d, ...>", a.nr16);   // (gap scores off the int16 grid: the strip kernels)
        pl.last_kernel = kn;
    }
    size_t budget = tb_budget_bytes();
    const bool semiglobal = mode >= 2;
    const size_t tb_elem_bytes = pl.split ? sizeof(uint2) : sizeof(uint4);
    const int lanes_per_task = pl.quad ? 16 : (pl.split ? 32 : 64);
    size_t t0 = 0;
    const size_t nt = pl.tasks.size();
    if (!getenv("PRALINE_TB_BUDGET_MB") && nt > 0) {
        // Plans of LONG sequences (more than 8 MiB of packed traceback per task: ~700 x 700 and up) run in chain mode
        // chunk by chunk; a chunk of a few tasks leaves the chip half empty, so they get up to 48 GiB (of 288) instead
        // of 8.  Measured, all pairs with paths: 256 x ~1000 aa 38 -> 32 ms, 128 x ~2500 96 -> 70 ms, 96 x ~5000
        // 371 -> 174 ms.  (Plans of many small tasks keep 8 GiB: within 3 % of 24 GiB on C3, see tb_budget_bytes.)
        int64_t all = 0;
        for (size_t t = 0; t < nt; ++t) all += pl.tb_elems[t] * (int64_t)tb_elem_bytes;
        if ((size_t)all > budget && all / (int64_t)nt > ((int64_t)8 << 20))
            budget = (size_t)std::min<int64_t>((int64_t)48 << 30, 2 * all);   // (twice: chunked plans cut at half the budget)
    }
    HIPCHK(hipEventRecord(pl.ev0, st));
    // ---- two passes with the PIPELINE as the forward fill (k_dp_pipe<..., KEEP>: operand rows streamed once per
    // workgroup, boundary hand-off through LDS, H recurrence) and k_trace_recompute on blocks of PRALINE_KEEP_BH rows:
    // float-profile arenas, global mode, no rectangles, plans whose scratch fits the budget (praline_plan_create)
    if (pl.pipe.ok && mode == PRALINE_MODE_GLOBAL && la.a16 != nullptr && !pl.has_rects &&
        !(getenv("PRALINE_TB_PIPE") && getenv("PRALINE_TB_PIPE")[0] == '0')) {
        char kn[160];
        snprintf(kn, sizeof(kn), "k_dp_pipe<%d, %d, false, false, true>", a.nr16, a.nterm16);
        pl.last_kernel = kn;
        if (pl.d_bnd2.n < (size_t)pl.pipe_keep_bnd_elems) RC(pl.d_bnd2.alloc((size_t)pl.pipe_keep_bnd_elems));
        if (pl.d_tb.n < (size_t)pl.pipe_keep_ck_floats * 4) RC(pl.d_tb.alloc((size_t)pl.pipe_keep_ck_floats * 4));
        PipeLaunch pp;
        pp.items = pl.d_pipe_items.p;
        pp.n_items = (unsigned)pl.pipe.items.size();
        pp.tasks = pl.d_pipe_tasks.p;
        pp.set_one = pl.d_pipe_set_one.p;
        pp.lane_pair = pl.d_pipe_lane_pair.p;
        pp.bnd = pl.d_pipe_bnd.p;
        pp.analytic = pl.d_pipe_analytic.p;
        pp.analytic_rows = pl.pipe_analytic_rows;
        pp.analytic_valid = pl.pipe_analytic_mode == mode && pl.pipe_analytic_go == la.rp.go1 && pl.pipe_analytic_ge == la.rp.ge1;
        pl.pipe_analytic_mode = mode; pl.pipe_analytic_go = la.rp.go1; pl.pipe_analytic_ge = la.rp.ge1;
        pp.scores = la.scores;
        pp.rp = la.rp;
        pp.stream = st;
        int rc2 = praline_launch_pipe_keep(pp, a16, a.nr16, a.nterm16, pl.d_bnd2.p, (float *)pl.d_tb.p, pl.d_end_cells.p,
                                           pl.d_pipe_analytic4.p);
        if (rc2 != PRALINE_OK) return fail(rc2, "no kept-state pipeline instance for nr=%d nterm=%d", a.nr16, a.nterm16);
        Trace2Args ta;
        ta.slot_off = pl.d_slot_off.p;
        ta.paths = pl.d_paths.p;
        ta.path_start = pl.d_path_start.p;
        ta.path_rows = pl.d_path_rows.p;
        LaunchArgs lb = la;
        lb.tasks = pl.d_pipe_tasks.p;
        lb.n_tasks = (unsigned)pl.pipe.tasks.size();
        lb.lane_one = pl.d_pipe_lane_one.p;
        lb.lane_pair = pl.d_pipe_lane_pair.p;
        lb.tb = (uint4 *)pl.d_tb.p;
        lb.bnd = pl.d_bnd2.p;
        rc2 = praline_launch_tb2_backward(lb, a16, ta, a.nr16, a.nterm16, false, false, 2, pl.d_pipe_analytic4.p);
        if (rc2 != PRALINE_OK) return fail(rc2, "no two-pass backward instance for nr=%d nterm=%d", a.nr16, a.nterm16);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(pl.ev1, st));
        return PRALINE_OK;
    }
    // ---- two passes (dp_trace2.hip.h) for plans too large for chain mode: a flag-free forward fill that keeps the
    // strip boundary columns and (M, U, L) of every 32nd row, then k_trace_recompute rebuilds the flags of only the
    // 32 x 32 blocks each path crosses.  PRALINE_TB_TWOPASS=0 keeps the single pass.
    {
        int64_t single_bytes = 0;
        int all_strips = 0;
        for (size_t t = 0; t < nt; ++t) { single_bytes += pl.tb_elems[t] * (int64_t)tb_elem_bytes; all_strips = std::max(all_strips, (int)pl.tasks[t].nstrips); }
        // (plans over the budget are cut into chunks of about half the budget, each of which can run in chain mode)
        const int64_t chunk_tasks = (size_t)single_bytes <= budget ? (int64_t)nt
                                                                   : (int64_t)((double)nt * (double)(budget / 2) / (double)single_bytes) + 1;
        const bool would_chain = pl.split && la.a16 != nullptr && all_strips >= 2 && chunk_tasks <= chain_max_tasks() &&
                                 !(getenv("PRALINE_NO_CHAIN") && getenv("PRALINE_NO_CHAIN")[0] == '1');
        // Default: LOCAL plans only.  Measured on C3 (1 047 552 alignments of ~250 aa, one-hot): local 60.1 -> 47.3 ms,
        // global 54.7 -> 53.7 ms (the forward fill's extra stores and a recompute of ~half the cells eat the saving when
        // every path spans the whole matrix).  PRALINE_TB_TWOPASS=1: every mode, =2: also instead of chain mode, =0: never.
        const char *tp = getenv("PRALINE_TB_TWOPASS");
        const int tpv = tp ? atoi(tp) : -1;
        // ---- two passes with the forward fill on the staged SCORES kernel (k_dp_split16<..., KEEP>: LDS-DMA operand
        // stream, shared-wave workgroups, 9 instead of ~20 VALU operations per cell) - float-profile arenas, global
        // mode, plans of one chunk.  PRALINE_TB_KEEP=1 enables it.
        {
            const char *kp = getenv("PRALINE_TB_KEEP");
            const int kpv = kp ? atoi(kp) : -1;
            bool keep = pl.split && la.a16 != nullptr && a16.stage && kpv != 0 && tpv != 0 && mode == PRALINE_MODE_GLOBAL &&
                        !pl.has_rects && a.nterm16 != 1 && (!pl.wg.empty() || !pl.wg_singles.empty()) &&
                        kpv == 1;   // opt-in while it is being tuned (C2: 5.8 ms against 5.9 in chain mode)
            if (getenv("PRALINE_DEBUG_KEEP"))
                fprintf(stderr, "keep=%d split=%d a16=%d stage=%d kpv=%d tpv=%d mode=%d rects=%d nterm=%d wg=%zu singles=%zu nt=%zu\n", (int)keep,
                        (int)pl.split, la.a16 != nullptr, a16.stage, kpv, tpv, mode, (int)pl.has_rects, a.nterm16, pl.wg.size(), pl.wg_singles.size(), nt);
            int64_t ck_e = 0, bnd_e = 0;   // floats, float4s
            if (keep) {
                for (size_t t = 0; t < nt; ++t) {
                    const WaveTask &wt = pl.tasks[t];
                    ck_e += (int64_t)wt.nstrips * PRALINE_TB2_CKPT_BLOCKS(wt.max_l1) * PRALINE_TB2_CKPT_FLOATS;
                    bnd_e += (int64_t)(wt.nstrips + 1) * (wt.max_l1 + PRALINE_TB2_PAD_ROWS) * 32;
                }
                keep = (size_t)(ck_e * 4 + bnd_e * 16) <= budget;
            }
            if (keep) {
                char kn[160];
                snprintf(kn, sizeof(kn), "k_dp_split16<%d, %d, false, 2, 4, true>", a.nr16, a.nterm16);
                pl.last_kernel = kn;
                ck_e = 0; bnd_e = 0;
                for (size_t t = 0; t < nt; ++t) {
                    WaveTask &wt = pl.tasks[t];
                    wt.tb_off = ck_e;
                    wt.aux_off = bnd_e;
                    ck_e += (int64_t)wt.nstrips * PRALINE_TB2_CKPT_BLOCKS(wt.max_l1) * PRALINE_TB2_CKPT_FLOATS;
                    bnd_e += (int64_t)(wt.nstrips + 1) * (wt.max_l1 + PRALINE_TB2_PAD_ROWS) * 32;
                }
                if (pl.d_tb.n < (size_t)ck_e * 4) RC(pl.d_tb.alloc((size_t)ck_e * 4));
                if (pl.d_bnd2.n < (size_t)bnd_e) RC(pl.d_bnd2.alloc((size_t)bnd_e));
                if (!pl.d_tasks.p) RC(pl.d_tasks.alloc(nt));
                HIPCHK(hipMemcpyAsync(pl.d_tasks.p, pl.tasks.data(), nt * sizeof(WaveTask), hipMemcpyHostToDevice, st));
                la.tasks = pl.d_tasks.p;
                la.n_tasks = (unsigned)nt;
                la.aux = nullptr;
                if (!pl.wg.empty()) {
                    if (!pl.d_wg.p) { RC(pl.d_wg.upload(pl.wg, st)); }
                    la.wg = pl.d_wg.p;
                    la.n_wg = (unsigned)pl.wg.size();
                } else {
                    if (!pl.d_wg_singles.p) { RC(pl.d_wg_singles.upload(pl.wg_singles, st)); }
                    la.wg = pl.d_wg_singles.p;
                    la.n_wg = (unsigned)pl.wg_singles.size();
                }
                int rc2 = praline_launch_keep_forward(la, a16, a.nr16, a.nterm16, pl.d_bnd2.p, (float *)pl.d_tb.p);
                if (rc2 != PRALINE_OK) return fail(rc2, "no kept-state forward instance for nr=%d nterm=%d", a.nr16, a.nterm16);
                Trace2Args ta;
                ta.slot_off = pl.d_slot_off.p;
                ta.paths = pl.d_paths.p;
                ta.path_start = pl.d_path_start.p;
                ta.path_rows = pl.d_path_rows.p;
                la.tb = (uint4 *)pl.d_tb.p;
                la.bnd = pl.d_bnd2.p;
                rc2 = praline_launch_tb2_backward(la, a16, ta, a.nr16, a.nterm16, false, false, 1);
                if (rc2 != PRALINE_OK) return fail(rc2, "no two-pass backward instance for nr=%d nterm=%d", a.nr16, a.nterm16);
                HIPCHK(hipGetLastError());
                HIPCHK(hipEventRecord(pl.ev1, st));
                return PRALINE_OK;
            }
        }
        const bool twopass = pl.split && !pl.quad && !pl.run_pk16 && la.a16 != nullptr && tpv != 0 && (tpv == 2 || (!would_chain && (local || tpv == 1)));
        if (twopass) {
            char kn[160];
            snprintf(kn, sizeof(kn), "k_dp_split16_tb<%d, %d, %s, %s, false, true>", a.nr16, tb_nterm, local ? "true" : "false",
                     pl.has_rects ? "true" : "false");
            pl.last_kernel = kn;
            if (pl.bnd_off0.size() != nt) { pl.bnd_off0.resize(nt); for (size_t t = 0; t < nt; ++t) pl.bnd_off0[t] = pl.tasks[t].bnd_off; }
            // the chunk cutting below rewrites the tasks' boundary offsets; the single pass and chain mode address the
            // plan's shared boundary buffer through the scheduler's offsets: put them back on EVERY way out
            struct RestoreBnd {
                praline_plan &pl;
                ~RestoreBnd() { for (size_t t = 0; t < pl.tasks.size() && t < pl.bnd_off0.size(); ++t) pl.tasks[t].bnd_off = pl.bnd_off0[t]; }
            } restore_bnd{pl};
            if (!pl.d_tasks.p) RC(pl.d_tasks.alloc(nt));
            int rc2 = PRALINE_OK;
            // plans that need several chunks: half the budget per chunk, two scratch sets, alternating streams
            size_t chunk_budget = budget;
            {
                int64_t all = 0;
                for (size_t t = 0; t < nt; ++t)
                    all += (int64_t)pl.tasks[t].nstrips * PRALINE_TB2_CKPT_BLOCKS(pl.tasks[t].max_l1) * PRALINE_TB2_CKPT_FLOATS * 4 +
                           (int64_t)(pl.tasks[t].nstrips + 1) * (pl.tasks[t].max_l1 + PRALINE_TB2_PAD_ROWS) * 32 * 16;
                if ((size_t)all > budget) chunk_budget = budget / 2;
            }
            // The chunks are cut first and each scratch set is allocated ONCE, for its largest chunk: a buffer that grew
            // in the middle of the loop would hand its old block back to the pool while the kernels of an earlier chunk
            // may still be using it - and the pool could give it to the OTHER set, which runs on the other stream
            // (seen with 45 000 alignments of ~1 000 x 1 300: thousands of wrong paths, different from run to run).
            struct Chunk2 { size_t t0, t1; int64_t ck_e, bnd_e, aux_e; };
            std::vector<Chunk2> chunks;
            while (t0 < nt) {
                size_t t1 = t0;
                int64_t ck_e = 0, bnd_e = 0, aux_e = 0;   // floats, float4s, floats
                while (t1 < nt) {
                    const WaveTask &wt = pl.tasks[t1];
                    const int64_t ck_add = (int64_t)wt.nstrips * PRALINE_TB2_CKPT_BLOCKS(wt.max_l1) * PRALINE_TB2_CKPT_FLOATS;
                    const int64_t bnd_add = (int64_t)(wt.nstrips + 1) * (wt.max_l1 + PRALINE_TB2_PAD_ROWS) * 32;
                    if (t1 > t0 && (size_t)((ck_e + ck_add) * 4 + (bnd_e + bnd_add) * 16) > chunk_budget) break;
                    pl.tasks[t1].tb_off = ck_e;
                    pl.tasks[t1].bnd_off = bnd_e;
                    pl.tasks[t1].aux_off = aux_e;
                    ck_e += ck_add;
                    bnd_e += bnd_add;
                    aux_e += semiglobal ? pl.aux_elems[t1] : 0;
                    ++t1;
                }
                chunks.push_back({t0, t1, ck_e, bnd_e, aux_e});
                t0 = t1;
            }
            {
                int64_t need_ck[2] = {0, 0}, need_bk[2] = {0, 0}, need_ax[2] = {1, 1};
                for (size_t c = 0; c < chunks.size(); ++c) {
                    need_ck[c & 1] = std::max(need_ck[c & 1], chunks[c].ck_e * 4);
                    need_bk[c & 1] = std::max(need_bk[c & 1], chunks[c].bnd_e);
                    need_ax[c & 1] = std::max(need_ax[c & 1], chunks[c].aux_e);
                }
                // (an earlier run's kernels may still be reading a block that is replaced here: the pool is stream-ordered -
                // the old block is not handed out again before both streams have passed this point - so no host wait)
                if (pl.d_tb.n < (size_t)need_ck[0]) RC(pl.d_tb.alloc((size_t)need_ck[0]));
                if (pl.d_bnd2.n < (size_t)need_bk[0]) RC(pl.d_bnd2.alloc((size_t)need_bk[0]));
                if (pl.d_aux.n < (size_t)need_ax[0]) RC(pl.d_aux.alloc((size_t)need_ax[0]));
                if (chunks.size() > 1) {
                    if (pl.d_tb_b.n < (size_t)need_ck[1]) RC(pl.d_tb_b.alloc((size_t)need_ck[1]));
                    if (pl.d_bnd2_b.n < (size_t)need_bk[1]) RC(pl.d_bnd2_b.alloc((size_t)need_bk[1]));
                    if (pl.d_aux_b.n < (size_t)need_ax[1]) RC(pl.d_aux_b.alloc((size_t)need_ax[1]));
                }
            }
            HIPCHK(hipMemcpyAsync(pl.d_tasks.p, pl.tasks.data(), nt * sizeof(WaveTask), hipMemcpyHostToDevice, st));   // (before the fork)
            bool forked = false;
            for (size_t c = 0; c < chunks.size() && rc2 == PRALINE_OK; ++c) {
                const int set = (int)(c & 1);
                hipStream_t cs = set ? g_rt.stream2 : st;
                if (set && !forked) {
                    if (hipEventRecord(g_rt.ev_fork, st) != hipSuccess || hipStreamWaitEvent(g_rt.stream2, g_rt.ev_fork, 0) != hipSuccess) {
                        rc2 = fail(PRALINE_ERR_DEVICE, "stream fork failed");
                        break;
                    }
                    forked = true;
                }
                DevBuf<char> &d_ck = set ? pl.d_tb_b : pl.d_tb;
                DevBuf<float4> &d_bk = set ? pl.d_bnd2_b : pl.d_bnd2;
                DevBuf<float> &d_ax = set ? pl.d_aux_b : pl.d_aux;
                la.stream = cs;
                const size_t c0 = chunks[c].t0, t1 = chunks[c].t1;
                la.tasks = pl.d_tasks.p + c0;
                la.lane_one = pl.d_lane_one.p + c0 * 32;
                la.lane_pair = pl.d_lane_pair.p + c0 * 32;
                la.tb = (uint4 *)d_ck.p;
                la.bnd = d_bk.p;
                la.aux = d_ax.p;
                la.n_tasks = (unsigned)(t1 - c0);
                rc2 = praline_launch_tb2_forward(la, a16, a.nr16, tb_nterm, local, pl.has_rects);
                if (rc2 != PRALINE_OK) { rc2 = fail(rc2, "no two-pass forward instance for nr=%d nterm=%d", a.nr16, tb_nterm); break; }
                if (semiglobal) {
                    const int64_t lanes = (int64_t)(t1 - c0) * 32;
                    hipLaunchKernelGGL(k_semiglobal_end, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, cs, la.ar, pl.d_tasks.p,
                                       pl.d_lane_one.p, pl.d_lane_pair.p, pl.d_pairs.p, d_ax.p, pl.d_end_cells.p, la.scores,
                                       la.rp, (int32_t)c0, (int32_t)t1, 1);
                }
                Trace2Args ta;
                ta.slot_off = pl.d_slot_off.p;
                ta.paths = pl.d_paths.p;
                ta.path_start = pl.d_path_start.p;
                ta.path_rows = pl.d_path_rows.p;
                rc2 = praline_launch_tb2_backward(la, a16, ta, a.nr16, tb_nterm, local, pl.has_rects);
                if (rc2 != PRALINE_OK) { rc2 = fail(rc2, "no two-pass backward instance for nr=%d nterm=%d", a.nr16, tb_nterm); break; }
                if (hipGetLastError() != hipSuccess) { rc2 = fail(PRALINE_ERR_DEVICE, "two-pass launch failed"); break; }
            }
            la.stream = st;
            if (forked && (hipEventRecord(g_rt.ev_join, g_rt.stream2) != hipSuccess || hipStreamWaitEvent(st, g_rt.ev_join, 0) != hipSuccess))
                rc2 = fail(PRALINE_ERR_DEVICE, "stream join failed");
            if (rc2 != PRALINE_OK) return rc2;
            // (d_tasks holds two-pass offsets now: the next single-pass run uploads its own)
            HIPCHK(hipEventRecord(pl.ev1, st));
            return PRALINE_OK;
        }
    }
    // plans that need several chunks: half the budget per chunk, two scratch sets, alternating streams (the traceback
    // and the tail of chunk k overlap the fill of chunk k + 1)
    size_t chunk_budget = budget;
    {
        int64_t all = 0;
        for (size_t t = 0; t < nt; ++t) all += pl.tb_elems[t] * (int64_t)tb_elem_bytes;
        if ((size_t)all > budget) chunk_budget = budget / 2;
    }
    // (cut first, allocate each scratch set once for its largest chunk - see the two-pass loop above)
    struct Chunk1 { size_t t0, t1; int64_t tb_e, aux_e; };
    std::vector<Chunk1> chunks;
    while (t0 < nt) {
        size_t t1 = t0;
        int64_t tb_e = 0, aux_e = 0;
        while (t1 < nt) {
            const int64_t add = pl.tb_elems[t1];
            if (t1 > t0 && (size_t)(tb_e + add) * tb_elem_bytes > chunk_budget) break;
            pl.tasks[t1].tb_off = tb_e;
            pl.tasks[t1].aux_off = aux_e;
            tb_e += add;
            aux_e += semiglobal ? pl.aux_elems[t1] : 0;
            ++t1;
        }
        chunks.push_back({t0, t1, tb_e, aux_e});
        t0 = t1;
    }
    {
        size_t need_tb[2] = {0, 0}, need_ax[2] = {1, 1};
        for (size_t c = 0; c < chunks.size(); ++c) {
            need_tb[c & 1] = std::max(need_tb[c & 1], (size_t)chunks[c].tb_e * tb_elem_bytes);
            need_ax[c & 1] = std::max(need_ax[c & 1], (size_t)chunks[c].aux_e);
        }
        // (blocks replaced here may still be read by an earlier run's kernels: stream-ordered pool, no host wait)
        if (pl.d_tb.n < need_tb[0]) RC(pl.d_tb.alloc(need_tb[0]));
        if (pl.d_aux.n < need_ax[0]) RC(pl.d_aux.alloc(need_ax[0]));
        if (chunks.size() > 1) {
            if (pl.d_tb_b.n < need_tb[1]) RC(pl.d_tb_b.alloc(need_tb[1]));
            if (pl.d_aux_b.n < need_ax[1]) RC(pl.d_aux_b.alloc(need_ax[1]));
        }
    }
    if (!pl.d_tasks.p) RC(pl.d_tasks.alloc(nt));
    HIPCHK(hipMemcpyAsync(pl.d_tasks.p, pl.tasks.data(), nt * sizeof(WaveTask), hipMemcpyHostToDevice, st));   // (before the fork)
    bool forked = false;
    // every way out of the loop below joins the second stream again (an error return would otherwise leave stream2's
    // kernels unordered against whatever the main stream does next with the plan's buffers)
    struct JoinGuard {
        bool &forked; hipStream_t st;
        ~JoinGuard()
        {
            if (forked && hipEventRecord(g_rt.ev_join, g_rt.stream2) == hipSuccess) (void)hipStreamWaitEvent(st, g_rt.ev_join, 0);
            forked = false;
        }
    } join_guard{forked, st};
    // Chain mode (one wave per task AND strip, pipelined across workgroups: dp_split16_tb.hip.h) for chunks of up to
    // chain_max_tasks() tasks - single alignments, the merge steps of the progressive MSA, C2-sized batches, and the
    // chunks of plans whose packed traceback exceeds the scratch budget (long sequences: 32 640 alignments of ~1000 x
    // ~1000 were 96 ms in task mode, three chunks of 380 waves each).  Chain chunks share one set of boundary columns
    // and flags: they all run on the main stream.
    bool chain_chunks = pl.split && !pl.quad && la.a16 != nullptr && !(getenv("PRALINE_NO_CHAIN") && getenv("PRALINE_NO_CHAIN")[0] == '1');
    {
        int64_t need_bnd = 0;
        size_t need_flags = 0;
        for (size_t c = 0; c < chunks.size() && chain_chunks; ++c) {
            int max_strips = 0;
            int64_t bnd_e = 0;
            for (size_t t = chunks[c].t0; t < chunks[c].t1; ++t) {
                max_strips = std::max(max_strips, (int)pl.tasks[t].nstrips);
                bnd_e += (int64_t)(pl.tasks[t].nstrips + 1) * (pl.tasks[t].max_l1 + 24) * 32;
            }
            if (max_strips < 2 || (int64_t)(chunks[c].t1 - chunks[c].t0) > chain_max_tasks()) chain_chunks = false;
            need_bnd = std::max(need_bnd, bnd_e);
            need_flags = std::max(need_flags, (chunks[c].t1 - chunks[c].t0) * (size_t)(max_strips + 1));
        }
        if (chain_chunks) {
            if (pl.d_bnd_chain.n < (size_t)need_bnd * sizeof(float4)) RC(pl.d_bnd_chain.alloc((size_t)need_bnd * sizeof(float4)));
            if (pl.d_chain_flags.n < need_flags) RC(pl.d_chain_flags.alloc(need_flags));
            if (local && pl.d_chain_cand.n < need_flags * 32) RC(pl.d_chain_cand.alloc(need_flags * 32));
        }
    }
    for (size_t c = 0; c < chunks.size(); ++c) {
        const int set = (int)(c & 1);
        hipStream_t cs = (set && !chain_chunks) ? g_rt.stream2 : st;
        if (set && !forked && !chain_chunks) {
            HIPCHK(hipEventRecord(g_rt.ev_fork, st));
            HIPCHK(hipStreamWaitEvent(g_rt.stream2, g_rt.ev_fork, 0));
            forked = true;
        }
        DevBuf<char> &d_tbs = set ? pl.d_tb_b : pl.d_tb;
        DevBuf<float> &d_ax = set ? pl.d_aux_b : pl.d_aux;
        la.stream = cs;
        t0 = chunks[c].t0;
        const size_t t1 = chunks[c].t1;
        la.tasks = pl.d_tasks.p + t0;
        la.lane_one = pl.d_lane_one.p + t0 * lanes_per_task;
        la.lane_pair = pl.d_lane_pair.p + t0 * lanes_per_task;
        la.tb = (uint4 *)d_tbs.p;
        la.aux = d_ax.p;
        la.n_tasks = (unsigned)(t1 - t0);
        int max_strips = 0;
        for (size_t t = t0; t < t1; ++t) max_strips = std::max(max_strips, (int)pl.tasks[t].nstrips);
        const bool chain = chain_chunks;
        if (chain) {
            const size_t nc = t1 - t0;   // tasks of this chunk
            std::vector<WaveTask> ct(pl.tasks.begin() + (std::ptrdiff_t)t0, pl.tasks.begin() + (std::ptrdiff_t)t1);
            int64_t bnd_e = 0;
            for (WaveTask &wt : ct) {
                wt.bnd_off = bnd_e;
                bnd_e += (int64_t)(wt.nstrips + 1) * (wt.max_l1 + 24) * 32;   // float4 elements, [strip boundary][row][32]
            }
            const size_t n_flags = nc * (size_t)(max_strips + 1);
            HIPCHK(hipMemsetAsync(pl.d_chain_flags.p, 0, n_flags * sizeof(int), st));
            HIPCHK(hipMemcpyAsync(pl.d_tasks.p + t0, ct.data(), nc * sizeof(WaveTask), hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));   // ct goes out of scope
            la.bnd = pl.d_bnd_chain.p;
            // rows between two publishes of a strip's progress: few for plans whose strip waves all run at once
            // (a single alignment: the next strip follows a few rows behind), many once a strip level alone
            // fills the chip (the consumers are dispatched a round later; every publish drains the stores)
            int every = nc >= 512 ? 96 : (nc >= 64 ? 24 : 6);
            {   // short sequences: at least four publishes per strip (co-resident consumers would wait for the end)
                int rows = 0;
                for (const WaveTask &wt : ct) rows = std::max(rows, (int)wt.max_l1);
                every = std::min(every, std::max(6, rows / 4));
            }
            // (k_dp_pk16_tb: two rows per step and shorter steps - measured on C2 one-hot 96 rows 1.55, 24 rows 1.60 TCUPS;
            // 2 016 pairs 12 rows; one alignment 6 rows: scripts/exp_pk16_chain.py)
            if (pl.run_pk16) every = std::min(every, nc >= 512 ? 24 : (nc >= 64 ? 12 : 6));
            if (const char *env = getenv("PRALINE_CHAIN_EVERY")) every = std::max(6, atoi(env));
            if (pl.run_pk16) {
                char kn[160];
                snprintf(kn, sizeof(kn), "k_dp_pk16_tb<%d, %s, %d, true>", a.nr16, local ? "true" : "false", pk16_slots);
                pl.last_kernel = kn;
            }
            int rc = pl.run_pk16 ? praline_launch_pk16_tb_chain(la, a16, a.nr16, local, pk16_slots, pk16_scale, max_strips, pl.d_chain_flags.p,
                                                                pl.d_chain_cand.p, every)
                                 : praline_launch_split16_tb_chain(la, a16, a.nr16, tb_nterm, local, pl.has_rects, max_strips,
                                                                   pl.d_chain_flags.p, pl.d_chain_cand.p, every);
            if (rc != PRALINE_OK) return fail(rc, "no chain instance of the path kernel for nr=%d nterm=%d", a.nr16, tb_nterm);
            if (local) {
                const int64_t lanes = (int64_t)nc * 32;
                hipLaunchKernelGGL(k_chain_local_end, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, la.tasks,
                                   la.lane_pair, pl.d_chain_cand.p, (int)nc, max_strips + 1, pl.d_end_cells.p, la.scores);
            }
            la.bnd = pl.d_bnd.p;
        } else if (pl.quad) {
            int rc = praline_launch_quad_tb(la, a16, a.nr16, tb_nterm == 1, local, pl.mask_kind);
            if (rc != PRALINE_OK) return fail(rc, "no k_dp_quad_tb instance for nr=%d", a.nr16);
        } else if (pl.run_pk16) {
            int rc = praline_launch_pk16_tb(la, a16, a.nr16, local, pk16_slots, pk16_scale);
            if (rc != PRALINE_OK) return fail(rc, "no k_dp_pk16_tb instance for nr=%d", a.nr16);
        } else {
            int rc = praline_launch_split16_tb(la, a16, a.nr16, tb_nterm, local, pl.has_rects);
            if (rc != PRALINE_OK) return fail(rc, "no k_dp_split16_tb instance for nr=%d nterm=%d", a.nr16, tb_nterm);
        }
        HIPCHK(hipGetLastError());
        RC(launch_traceback(pl, la, t0, t1, mode));
    }
    la.stream = st;
    if (forked) {
        HIPCHK(hipEventRecord(g_rt.ev_join, g_rt.stream2));
        HIPCHK(hipStreamWaitEvent(st, g_rt.ev_join, 0));
        forked = false;
    }
    HIPCHK(hipEventRecord(pl.ev1, st));
    return PRALINE_OK;
}

// praline_plan_run with the arena's per-position gap scores (praline_arena_set_gap_scores) instead of one (open, extend):
// U[y][x] takes the scores of position y - 1 of sequence one, L[y][x] those of position x - 1 of sequence two
// (cext.c:155-158,172-175), the boundary cells follow align.py:371-385.  The plan must have been created while the
// arena held gap scores (such plans read their match scores from dense tiles, plan_run_dense).
extern "C" int praline_plan_run_gaps(praline_plan *plan, int mode, void *d_scores)
{
    if (!plan) return fail(PRALINE_ERR_ARG, "plan is NULL");
    if (!plan->ppg) return fail(PRALINE_ERR_UNSUPPORTED, "the plan was created before praline_arena_set_gap_scores");
    if (!plan->arena->has_gaps || !plan->arena->d_gaps.p) return fail(PRALINE_ERR_ARG, "the arena holds no gap scores");
    plan->run_ppg = true;
    const int rc = praline_plan_run(plan, mode, 0.0f, 0.0f, d_scores);
    plan->run_ppg = false;
    return rc;
}

extern "C" int praline_plan_kernel_name(const praline_plan *plan, char *buf, int64_t size)
{
    if (!plan || !buf || size <= 0) return fail(PRALINE_ERR_ARG, "NULL argument");
    snprintf(buf, (size_t)size, "%s", plan->last_kernel.c_str());
    return PRALINE_OK;
}

extern "C" int praline_plan_kernel_resources(const praline_plan *plan, int32_t *vgprs, int32_t *lds_bytes, int32_t *waves_per_simd)
{
    if (!plan || !vgprs || !lds_bytes || !waves_per_simd) return fail(PRALINE_ERR_ARG, "NULL argument");
    *vgprs = *lds_bytes = *waves_per_simd = 0;
    if (!plan->pipe.ok || plan->last_mode < 0) return PRALINE_OK;   // (reported for the pipeline workgroups only)
    int v = 0, l = 0;
    if (plan->want_paths) {
        // (path plans: the pipeline is the forward fill of global runs only)
        if (plan->last_mode != PRALINE_MODE_GLOBAL || plan->last_kernel.compare(0, 9, "k_dp_pipe") != 0) return PRALINE_OK;
        RC(praline_pipe_keep_attrs(plan->arena->nr16, plan->arena->nterm16, &v, &l));
    } else
    RC(praline_pipe_attrs(plan->arena->nr16, plan->arena->nterm16, plan->last_mode, &v, &l));
    *vgprs = v;
    *lds_bytes = l;
    // MI355X_MICROARCH.md, register files: allocation granule 8, 512 registers per lane and SIMD; 160 KiB of LDS per CU;
    // a workgroup of four waves puts one wave on every SIMD
    const int by_regs = std::min(8, 512 / std::max(8, (v + 7) / 8 * 8));
    const int by_lds = l > 0 ? (160 * 1024) / l : 8;
    *waves_per_simd = std::min(by_regs, by_lds);
    return PRALINE_OK;
}

extern "C" int praline_plan_last_timing(praline_plan *plan, float *kernel_ms)
{
    if (!plan || !kernel_ms) return fail(PRALINE_ERR_ARG, "NULL argument");
    if (plan->n_pairs == 0) { *kernel_ms = 0.0f; return PRALINE_OK; }
    if (plan->last_mode < 0) return fail(PRALINE_ERR_ARG, "praline_plan_run has not been called");
    HIPCHK(hipEventSynchronize(plan->ev1));
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, plan->ev0, plan->ev1));
    plan->last_kernel_ms = ms;
    *kernel_ms = ms;
    return PRALINE_OK;
}

extern "C" int praline_plan_scores(praline_plan *plan, float *scores)
{
    if (!plan || (!scores && plan->n_pairs)) return fail(PRALINE_ERR_ARG, "NULL argument");
    if (plan->n_pairs == 0) return PRALINE_OK;
    if (!plan->last_scores) return fail(PRALINE_ERR_ARG, "praline_plan_run has not been called");
    // the buffer the last run wrote: the plan's own or the caller's d_scores
    HIPCHK(hipMemcpyAsync(scores, plan->last_scores, (size_t)plan->n_pairs * sizeof(float), hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    return PRALINE_OK;
}

extern "C" int praline_plan_paths(praline_plan *plan, int32_t *paths, int64_t *path_off, int32_t *path_rows)
{
    if (!plan || !paths || !path_off || !path_rows) return fail(PRALINE_ERR_ARG, "NULL argument");
    if (!plan->want_paths) return fail(PRALINE_ERR_ARG, "plan was created without want_paths");
    if (plan->n_pairs == 0) return PRALINE_OK;
    hipStream_t st = g_rt.stream;
    HIPCHK(hipMemcpyAsync(paths, plan->d_paths.p, (size_t)plan->path_cap * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(path_off, plan->d_path_start.p, (size_t)plan->n_pairs * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(path_rows, plan->d_path_rows.p, (size_t)plan->n_pairs * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return PRALINE_OK;
}
