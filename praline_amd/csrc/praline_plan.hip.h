// praline_plan.hip.h -- part of praline_dp.hip (one translation unit; included there, in this order): the pair plan: its device state and praline_plan_create (host scheduling in sched.cpp, uploads).
// --------------------------------------------------------------------------------------------
// plan
// --------------------------------------------------------------------------------------------
struct praline_plan {
    praline_arena *arena = nullptr;
    int64_t n_pairs = 0;
    int64_t cells = 0;
    int64_t path_cap = 0;
    bool want_paths = false;
    bool has_rects = false;
    int max_rects = 0;    // rectangles of the pair with the most (lists given at creation)
    int64_t count_runs = -1;            // praline_plan_add_counts: runs of pairs with one master (-1: not looked for yet, 0: none)
    DevBuf<int64_t> d_count_runs;
    int slot_rects = -1;  // >= 0: the rectangles live in fixed slots on the device (praline_plan_mask_path_bounds), this many used
    int mask_kind = 0;   // 0 none, 1 <= PRALINE_MAX_RECTS rectangles per pair (registers), 2 any number (per-row mask words, k_build_zmask)
    int tp = 1;
    bool split = false;  // k_dp_split task layout
    bool quad = false;   // path plan on a one-hot arena in the 16-pairs-per-task layout of k_dp_quad_tb (dp_quad.hip.h)
    // path plan on a one-hot arena with an integral exchange matrix whose DP values fit int16: the 32-pair layout with planes
    // sized for k_dp_pk16_tb (dp_pk16.hip.h); a run whose gap scores do not qualify takes the strip kernels on the same tasks
    bool pk16 = false;
    bool run_pk16 = false;   // the run in progress / the last run used k_dp_pk16_tb
    std::vector<WaveTask> tasks;
    std::vector<int64_t> tb_elems;  // per task, uint4 elements
    std::vector<int64_t> aux_elems; // per task, floats
    int64_t bnd_elems = 0;
    DevBuf<WaveTask> d_tasks;
    DevBuf<WaveTask> d_tasks_chain;   // scores-only chain mode: the tasks with chain-mode boundary offsets
    int scores_chain = -1;            // -1: not decided, 0 / 1: this score plan runs in chain mode (plan_scores_chain_wanted)
    // small batches: four-wave workgroups whose waves share long tasks (k_dp_split16 WPG = 4, WgDesc)
    std::vector<WgDesc> wg;
    DevBuf<WgDesc> d_wg;
    // large batches in LOCAL mode: the same kernel with four independent tasks per workgroup (its one-wave LOCAL
    // instances need 256 VGPRs + ~130 AGPRs, the four-wave ones 185-219: two waves per SIMD)
    std::vector<WgDesc> wg_singles;
    DevBuf<WgDesc> d_wg_singles;
    // pipeline workgroups (k_dp_pipe, dp_pipe.hip.h): scores-only plans on float-profile arenas
    PipeSchedule pipe;
    DevBuf<PipeItem> d_pipe_items;
    DevBuf<WaveTask> d_pipe_tasks;
    DevBuf<int32_t> d_pipe_set_one, d_pipe_lane_pair;
    DevBuf<float2> d_pipe_bnd, d_pipe_analytic;
    // path plans (global mode): the pipeline as the forward fill of the two-pass scheme (k_dp_pipe<..., KEEP> +
    // k_trace_recompute): per-task sequences one, the float4 analytic column, scratch sizes (kept columns in d_bnd2, row
    // checkpoints in d_tb); the tasks carry aux_off / tb_off into them
    DevBuf<int32_t> d_pipe_lane_one;
    DevBuf<float4> d_pipe_analytic4;
    int64_t pipe_keep_bnd_elems = 0, pipe_keep_ck_floats = 0;
    int pipe_analytic_rows = 0;
    int pipe_analytic_mode = -1;            // mode and gap scores the analytic column was last written for (-1: never)
    float pipe_analytic_go = 0.0f, pipe_analytic_ge = 0.0f;
    DevBuf<int32_t> d_lane_one, d_lane_pair, d_pairs, d_rect_off, d_rects, d_end_cells, d_path_rows, d_paths;
    DevBuf<PairLoc> d_loc;
    DevBuf<float> d_scores, d_aux;
    DevBuf<char> d_bnd;
    DevBuf<float4> d_bnd2;      // two-pass mode: every strip's boundary column, kept for the recompute kernel
    std::vector<int64_t> bnd_off0;
    DevBuf<char> d_bnd_chain;   // chain mode: one boundary column per strip boundary
    DevBuf<int> d_chain_flags;  // chain mode: rows published per (task, strip)
    DevBuf<float4> d_chain_cand;  // chain mode, local: first-argmax candidate per (task, strip, pair)
    DevBuf<char> d_tb;
    // second scratch set of chunked path plans (chunks alternate between two streams)
    DevBuf<char> d_tb_b;
    DevBuf<float> d_aux_b;
    DevBuf<float4> d_bnd2_b;
    DevBuf<int64_t> d_slot_off, d_path_start;
    RawVec<int64_t> slot_off;
    float last_kernel_ms = 0.0f;
    int last_mode = -1;
    // plans whose DP reads its match scores from DENSE TILES (the dense-tile instances of k_dp_split16 / k_dp_split16_tb,
    // plan_run_dense) and who writes the tiles:
    //   1  k_match_tile - the reference's summation order (PRALINE_MATCH_REFERENCE) on arenas of up to 32 symbols whose rows
    //      hold at most 8 nonzeros (dp_reftile.hip.h)
    //   2  k_match_reft / k_match_ref, one cell per thread - the reference's order for every other arena (more than 32 active
    //      symbols, denser rows) and for plans with more than PRALINE_MAX_RECTS rectangles per pair on float profiles
    //   3  k_scores_tile_batch, the fp32 MFMA chain - plans created on an arena with per-position gap scores in the default
    //      match mode (both their constant-gap and their per-position runs)
    int dense_kind = 0;
    std::vector<int32_t> h_lane_pair, h_pairs;
    DevBuf<int32_t> d_chunk_pairs;
    bool ppg = false;       // created on an arena with per-position gap scores
    bool run_ppg = false;   // the run in progress uses them (praline_plan_run_gaps)
    DevBuf<float> d_dense;                  // the tiles of one launch chunk
    DevBuf<int64_t> d_dense_off;
    DevBuf<RefTileBlock> d_tile_blocks;
    DevBuf<int32_t> d_tile_grp;
    std::vector<int32_t> h_lane_one;        // [task][32], host copy (groups of tasks with the same sequences one)
    // mask_kind 2: column masks per (pair, strip, row) (k_build_zmask)
    DevBuf<unsigned> d_zmask;
    DevBuf<int64_t> d_zm_off;
    std::string last_kernel;        // the DP kernel instance the last run launched (as rocprofv3 names it)
    float *last_scores = nullptr;   // where the last praline_plan_run wrote the scores (own buffer or the caller's)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the last run's launches, on the launch stream
    ~praline_plan()
    {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
};

// chain mode (one wave per task and strip) for plans of up to this many tasks.  Measured with paths, float
// profiles, ms per run task mode -> chain mode: 120 pairs 6.6 -> 0.93, 2 016 pairs 6.6 -> 1.5, 8 128 pairs
// 7.1 -> 3.0, C2 (32 640 pairs, 1 144 tasks) 9.6 -> 7.5, 2 048 tasks 5.2 -> 5.2, 4 600 tasks 10.6 -> 11.5.
// Blocks are dispatched in index order and a strip's producer has the smaller index, so a chain never waits for
// a wave that has not been dispatched, whatever fits on the chip at once.
static int64_t chain_max_tasks()
{
    if (const char *env = getenv("PRALINE_CHAIN_MAX_TASKS")) return atoll(env);
    return 2304;   // measured crossover with task mode (scripts/exp_chain.py); within +-5 % of it up to ~4000 tasks
}

#define PRALINE_TB2_PAD_ROWS PRALINE_TB2_PAD

// traceback scratch budget per launch chunk (bytes)
static size_t tb_budget_bytes()
{
    if (const char *env = getenv("PRALINE_TB_BUDGET_MB")) return (size_t)atoll(env) << 20;
    // 8 GiB (two sets of 4 GiB once a plan needs several chunks): with the chunks alternating between two streams the
    // rate is within 3 % of a 24 GiB budget (C3), and a first-use hipMalloc of the scratch costs 0.2 s instead of 0.8
    return (size_t)8 << 30;
}

static size_t reftile_budget_bytes();

// scheduler options of the pipeline workgroups for a pair list of this size (plan creation and praline_sched_prepare)
static PipeOptions pipe_options_for(int64_t n_pairs)
{
    PipeOptions po;
    // sequences two per scheduler block: 32 while the whole plan is resident at once (up to ~2.5 tasks per workgroup
    // slot: C2 1.90 ms against 2.09 with 16), 16 beyond (one rank's share of C4: 47.6 ms against 48.8 with 32 - the
    // unions of 32 columns' sequences one leave more half-filled sets; scripts/exp_pipe_block2.py, exp_c4_block.py)
    po.block_twos = n_pairs <= 40000 ? 32 : 16;
    if (const char *env = getenv("PRALINE_PIPE_BLOCK")) po.block_twos = atoi(env);
    if (const char *env = getenv("PRALINE_PIPE_SLOTS")) po.wg_slots = atoll(env);
    return po;
}

// the pipeline schedule of a scores-only plan over `pairs` (everything praline_plan_create derives from the pair list and
// the sequence lengths alone); below ~200 tasks (all pairs of ~110 sequences) the shared-wave task schedule is as fast
// or faster (scripts/exp_pipe_sweep.py): a pipeline item cannot be smaller than one task
static void pipe_schedule_for(const int32_t *lens, int64_t n_seqs, int64_t n_pairs, const int32_t *pairs, int max_len, PipeSchedule &pipe)
{
    int min_len = max_len;
    (void)max_len;
    PhaseTimer pt("pipe_schedule_for");
    sched_pair_stats(lens, n_seqs, n_pairs, pairs, nullptr, &min_len, nullptr);
    pt.mark("shortest sequence of the list");
    if (min_len >= 1) build_pipe_schedule(lens, n_seqs, n_pairs, pairs, pipe_options_for(n_pairs), pipe);
    pt.mark("build_pipe_schedule");
    int64_t min_tasks = 200;
    if (const char *env = getenv("PRALINE_PIPE_MIN_TASKS")) min_tasks = atoll(env);
    if (pipe.ok && (int64_t)pipe.tasks.size() < min_tasks) pipe = PipeSchedule();
}

// praline_sched_prepare: host-only, may run on another host thread while the arena of the same sequences is created
struct praline_sched {
    std::vector<int32_t> lens;
    int64_t n_pairs = 0;
    std::vector<int32_t> pairs;   // the pair list the schedule belongs to (compared entry by entry when the plan is created)
    PipeSchedule pipe;
};

extern "C" int praline_sched_prepare(int64_t n_seqs, const int32_t *lens, int64_t n_pairs, const int32_t *pairs, praline_sched **out)
{
    if (!out) return fail(PRALINE_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n_seqs <= 0 || !lens || n_pairs < 0 || (n_pairs > 0 && !pairs)) return fail(PRALINE_ERR_ARG, "bad schedule arguments");
    if (n_pairs > (int64_t)INT32_MAX) return fail(PRALINE_ERR_ARG, "a schedule holds at most 2^31 - 1 pairs (split the list)");
    int max_len = 0;
    for (int64_t s = 0; s < n_seqs; ++s) {
        if (lens[s] <= 0) return fail(PRALINE_ERR_ARG, "sequence %lld has length %d (must be >= 1)", (long long)s, lens[s]);
        max_len = std::max(max_len, lens[s]);
    }
    {
        int64_t bad = -1;
        sched_pair_stats(lens, n_seqs, n_pairs, pairs, nullptr, nullptr, &bad);
        if (bad >= 0) return fail(PRALINE_ERR_ARG, "pair %lld = (%d, %d) out of range", (long long)bad, pairs[2 * bad], pairs[2 * bad + 1]);
    }
    praline_sched *sc = new praline_sched();
    sc->lens.assign(lens, lens + n_seqs);
    sc->n_pairs = n_pairs;
    if (n_pairs > 0) {
        sc->pairs.assign(pairs, pairs + 2 * n_pairs);
        pipe_schedule_for(lens, n_seqs, n_pairs, pairs, max_len, sc->pipe);
    }
    *out = sc;
    return PRALINE_OK;
}

extern "C" int praline_sched_destroy(praline_sched *sched)
{
    delete sched;
    return PRALINE_OK;
}

static int plan_create_impl(praline_arena *arena, int64_t n_pairs, const int32_t *pairs, int want_paths, const int32_t *rect_off,
                            const int32_t *rects, praline_sched *prep, praline_plan **out);

extern "C" int praline_plan_create(praline_arena *arena, int64_t n_pairs, const int32_t *pairs, int want_paths,
                                   const int32_t *rect_off, const int32_t *rects, praline_plan **out)
{
    return plan_create_impl(arena, n_pairs, pairs, want_paths, rect_off, rects, nullptr, out);
}

extern "C" int praline_plan_create_prepared(praline_arena *arena, int64_t n_pairs, const int32_t *pairs, praline_sched *sched,
                                            praline_plan **out)
{
    return plan_create_impl(arena, n_pairs, pairs, 0, nullptr, nullptr, sched, out);
}

static int plan_create_impl(praline_arena *arena, int64_t n_pairs, const int32_t *pairs, int want_paths, const int32_t *rect_off,
                            const int32_t *rects, praline_sched *prep, praline_plan **out)
{
    if (!out) return fail(PRALINE_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!arena || n_pairs < 0 || (n_pairs > 0 && !pairs)) return fail(PRALINE_ERR_ARG, "bad plan arguments");
    if (n_pairs > (int64_t)INT32_MAX) return fail(PRALINE_ERR_ARG, "a plan holds at most 2^31 - 1 pairs (split the list)");
    RC(arena_ready(arena));
    if ((rect_off != nullptr) != (rects != nullptr) && rect_off && rect_off[n_pairs] > 0)
        return fail(PRALINE_ERR_ARG, "rect_off given without rects");
    if (rect_off && !want_paths && rect_off[n_pairs] > 0)
        return fail(PRALINE_ERR_UNSUPPORTED, "zero rectangles are only supported together with want_paths");
    RC(ensure_runtime(-1));
    PhaseTimer pt("plan_create");
    const praline_arena &a = *arena;
    bool many_rects = false;   // some pair carries more rectangles than the register-resident mask code holds
    int max_rects = 0;         // the longest rectangle list of a pair
    int64_t list_cells = 0;
    {
        int64_t bad = -1;
        sched_pair_stats(a.len.data(), a.n_seqs, n_pairs, pairs, &list_cells, nullptr, &bad);
        if (bad >= 0) return fail(PRALINE_ERR_ARG, "pair %lld = (%d, %d) out of range", (long long)bad, pairs[2 * bad], pairs[2 * bad + 1]);
    }
    if (rect_off)
        for (int64_t p = 0; p < n_pairs; ++p) {
            if (rect_off[p + 1] < rect_off[p]) return fail(PRALINE_ERR_ARG, "rect_off is not ascending at pair %lld", (long long)p);
            many_rects = many_rects || rect_off[p + 1] - rect_off[p] > PRALINE_MAX_RECTS;
            max_rects = std::max(max_rects, (int)(rect_off[p + 1] - rect_off[p]));
        }
    pt.mark("pair list checked");
    praline_plan *pl = new praline_plan();
    pl->arena = arena;
    pl->n_pairs = n_pairs;
    pl->want_paths = want_paths != 0;
    pl->has_rects = rect_off && rect_off[n_pairs] > 0;
    pl->max_rects = max_rects;
    pl->mask_kind = !pl->has_rects ? 0 : (many_rects ? 2 : 1);

    // ---- host scheduling (sched.cpp): tasks, launch order, workgroup descriptors ----
    SchedOptions opt;
    opt.want_paths = pl->want_paths;
    // every plan runs on the split-strip layout (32 pairs per wave, both halves on the same pairs)
    opt.split_layout = true;
    // the strip kernels hold PRALINE_MAX_RECTS rectangles per pair in registers; plans with more per pair (many
    // Waterman-Eggert iterations: rare) read per-row column masks (k_build_zmask): k_dp_quad_tb for plain sequences, the
    // dense-tile instances for every other arena
    const bool quad_ok = want_paths && a.nr16 > 0 && a.nterm16 == 1 && a.onehot && match_mode() == PRALINE_MATCH_FAST &&
                         !(getenv("PRALINE_TB_QUAD") && getenv("PRALINE_TB_QUAD")[0] == '0');
    pl->ppg = a.has_gaps;
    // who forms the match scores (praline_plan::dense_kind): the reference's order on request, for arenas without packed
    // operands (more than 32 active symbols) and for many-rectangle plans on float profiles
    if (match_mode() == PRALINE_MATCH_REFERENCE || a.wide || (many_rects && !quad_ok)) {
        pl->dense_kind = 2;
        // arenas of up to 32 symbols whose rows hold at most 8 nonzeros: k_match_tile (PRALINE_NO_REFTILE=1: the
        // one-cell-per-thread kernels, as for the other arenas - the independent second implementation the tests compare with)
        if (match_mode() == PRALINE_MATCH_REFERENCE && !a.wide && a.nr16 > 0 &&
            !(getenv("PRALINE_NO_REFTILE") && getenv("PRALINE_NO_REFTILE")[0] == '1')) {
            int rc = arena_ensure_reft2(arena);
            if (rc != PRALINE_OK) { delete pl; return rc; }
            if (a.reft2_state == 1) pl->dense_kind = 1;
        }
    } else if (pl->ppg) {
        pl->dense_kind = 3;
    }
    // alignments with paths of plain sequences (exact-mode arenas with their symbol stream): k_dp_quad_tb, 16 pairs per
    // task (PRALINE_TB_QUAD=0: the 32-pair strip kernels, as for every other arena)
    {
        const char *tq = getenv("PRALINE_TB_QUAD");
        const Arena16Dev v16q = a.view16();
        const bool quad_kind = want_paths && pl->dense_kind == 0 && a.nr16 > 0 && a.nterm16 == 1 &&
                               v16q.sym8 != nullptr && match_mode() == PRALINE_MATCH_FAST && !(tq && tq[0] == '0');
        pl->quad = quad_kind;
        // integer scoring within int16 (the exchange matrix alone is checked here, the gap scores by every run): two pairs per
        // lane, k_dp_pk16_tb (PRALINE_TB_PK16=0: never).  Plans with more than PRALINE_MAX_RECTS rectangles per pair keep
        // k_dp_quad_tb, which reads mask words.
        const char *tk16 = getenv("PRALINE_TB_PK16");
        // (plans that fill the chip: a task is one wave and holds twice the pairs of a k_dp_quad_tb task - measured on C2,
        // 32 640 pairs = 1 020 such tasks on 1 024 SIMDs: 0.99 against 1.04 TCUPS; on a C3 slice of 130 944 pairs 1.74 against
        // 1.41.  PRALINE_TB_PK16=1: every plan that qualifies)
        // (smaller plans run it in chain mode, one wave per task and strip)
        pl->pk16 = pl->quad && !many_rects && a.all_onehot && a.s_scale_bits >= 0 && a.s_scale_bits <= 8 && !(tk16 && tk16[0] == '0') &&
                   (2.0 * a.max_len + 36.0) * (double)a.s_absmax * (double)(1 << a.s_scale_bits) < 32000.0;   // (+ 36: the boundary cells of a last strip's padding columns)
        if (pl->pk16) pl->quad = false;
        // k_dp_quad_tb has no chain mode: a task is one wave from the first strip to the last.  Plans that do not fill the chip
        // with such waves (measured: one alignment of 1 400 x 1 400 26 ms against 2 ms in chain mode; 2 016 pairs of ~400 3.8
        // against 1.1 ms; C2-sized plans level) keep the 32-pair strip kernels and their chain mode - unless the plan needs the
        // mask words only k_dp_quad_tb reads (PRALINE_TB_QUAD=1: always)
        if (pl->quad && n_pairs < 32768 && !many_rects && !(tq && tq[0] == '1')) pl->quad = false;
        opt.pk16 = pl->pk16;
        opt.quad16 = pl->quad;
    }
    if (const char *env = getenv("PRALINE_XCD_GROUP")) opt.xcd_group = atoi(env);
    if (const char *env = getenv("PRALINE_NO_W2")) opt.shared_waves = env[0] != '1';
    {   // score plans on one-hot arenas run the lookup instances: three waves per SIMD (168 VGPRs, 4.75 KB of LDS per wave)
        const char *nl = getenv("PRALINE_NO_LOOKUP");
        if (!want_paths && a.onehot && a.nterm16 == 1 && a.nr16 > 0 && match_mode() == PRALINE_MATCH_FAST && !(nl && nl[0] == '1'))
            opt.wave_slots = 3072;
    }
    if (const char *env = getenv("PRALINE_W_SLOTS")) opt.wave_slots = atoll(env);
    if (const char *env = getenv("PRALINE_W_SNAKE")) opt.snake = atoi(env) != 0;
    if (const char *env = getenv("PRALINE_WG_XCD")) opt.wg_xcd = atoi(env) != 0;
    if (const char *env = getenv("PRALINE_WG_BALANCE")) opt.balance = atoi(env) != 0;
    Schedule sch;
    // scores-only plans on float-profile arenas (128-byte operand rows): pipeline workgroups (PRALINE_NO_PIPE=1: the task
    // schedule above, as for every other kind of plan)
    {
        const char *np = getenv("PRALINE_NO_PIPE");
        const Arena16Dev v16 = a.view16();
        // path plans without rectangles get the pipeline schedule BESIDE their task schedule: global runs take it as the
        // forward fill of the two-pass scheme (PRALINE_TB_PIPE=0: never), the other modes keep chain / task mode
        const char *tpp = getenv("PRALINE_TB_PIPE");
        const bool paths_ok = !want_paths || (!pl->has_rects && !(tpp && tpp[0] == '0'));
        if (paths_ok && pl->dense_kind == 0 && a.nr16 > 0 && v16.stage && v16.sym8 == nullptr &&
            praline_pipe_supported(a.nr16, a.nterm16) && match_mode() == PRALINE_MATCH_FAST && !(np && np[0] == '1') && n_pairs > 0) {
            // (a schedule prepared from the same lengths and pair list while the arena was being created: take it)
            const bool prepared = prep != nullptr && prep->n_pairs == n_pairs && prep->lens == a.len &&
                                  memcmp(prep->pairs.data(), pairs, (size_t)n_pairs * 2 * sizeof(int32_t)) == 0;
            if (prepared) pl->pipe = std::move(prep->pipe);
            else pipe_schedule_for(a.len.data(), a.n_seqs, n_pairs, pairs, a.max_len, pl->pipe);
            if (prepared) { prep->pipe = PipeSchedule(); prep->n_pairs = -1; }   // (consumed)
        }
    }
    pt.mark("pipeline schedule");
    if (pl->pipe.ok && want_paths) {
        // scratch of the KEEP forward fill: per task (nstrips + 1) kept columns of max_l1 + PRALINE_TB2_PAD rows and
        // nstrips x pipe_keep_blocks row checkpoints; plans beyond the scratch budget keep chain / task mode
        int64_t bnd_e = 0, ck_e = 0;
        for (WaveTask &wt : pl->pipe.tasks) {
            wt.aux_off = bnd_e;
            wt.tb_off = ck_e;
            bnd_e += (int64_t)(wt.nstrips + 1) * (wt.max_l1 + PRALINE_TB2_PAD_ROWS) * 32;
            const int rows_top = std::max(wt.max_l1 + 12, PRALINE_PIPE_MIN_STEPS);
            ck_e += (int64_t)wt.nstrips * (rows_top / PRALINE_KEEP_BH + 1) * PRALINE_TB2_CKPT_FLOATS;
        }
        if ((size_t)(bnd_e * 16 + ck_e * 4) > tb_budget_bytes()) pl->pipe = PipeSchedule();
        else { pl->pipe_keep_bnd_elems = bnd_e; pl->pipe_keep_ck_floats = ck_e; }
    }
    if (pl->pipe.ok && !want_paths) {
        // the pipeline schedule is all a scores-only run needs: no task schedule, no per-task boundary scratch
        sch.split = opt.split_layout;
        sch.cells = list_cells;
    } else {
        build_schedule(a.len.data(), n_pairs, pairs, opt, sch);
    }
    pt.mark("host scheduling");
    pl->tp = sch.tp;
    pl->split = sch.split;
    pl->tasks.swap(sch.tasks);
    pl->tb_elems.swap(sch.tb_elems);
    pl->aux_elems.swap(sch.aux_elems);
    pl->bnd_elems = sch.bnd_elems;
    pl->wg.swap(sch.wg);
    pl->wg_singles.swap(sch.wg_singles);
    pl->slot_off.swap(sch.slot_off);
    pl->path_cap = sch.path_cap;
    pl->cells = sch.cells;
    const RawVec<int32_t> &lane_one = sch.lane_one, &lane_pair = sch.lane_pair;
    const RawVec<PairLoc> &loc = sch.loc;
    if (pl->dense_kind != 0) {
        pl->h_lane_pair.assign(sch.lane_pair.begin(), sch.lane_pair.end());
        pl->h_pairs.assign(pairs, pairs + 2 * n_pairs);
    }
    if (pl->dense_kind == 1) pl->h_lane_one.assign(sch.lane_one.begin(), sch.lane_one.end());
    if (!want_paths && !pl->pipe.ok && pl->h_pairs.empty() && (int64_t)pl->tasks.size() <= chain_max_tasks())
        pl->h_pairs.assign(pairs, pairs + 2 * n_pairs);   // (score plans that may run in chain mode: k_semiglobal_end reads the pairs)
    const int64_t bnd = pl->bnd_elems, cap = pl->path_cap;

    hipStream_t st = g_rt.stream;
    int rc = PRALINE_OK;
    if ((rc = pl->d_lane_one.upload(lane_one, st)) || (rc = pl->d_lane_pair.upload(lane_pair, st)) ||
        (rc = pl->d_scores.alloc((size_t)n_pairs)) ||
        (rc = pl->d_bnd.alloc((size_t)bnd * (want_paths ? sizeof(float4) : sizeof(float2))))) {
        delete pl;
        return rc;
    }
    if (pl->dense_kind != 0 && !want_paths) {   // (the per-cell match-score kernels and k_semiglobal_end read them)
        if ((rc = pl->d_pairs.upload(pl->h_pairs, st)) || (rc = pl->d_loc.upload(loc, st))) { delete pl; return rc; }
    }
    if (pl->pipe.ok) {
        if ((rc = pl->d_pipe_items.upload(pl->pipe.items, st)) || (rc = pl->d_pipe_tasks.upload(pl->pipe.tasks, st)) ||
            (rc = pl->d_pipe_set_one.upload(pl->pipe.set_one, st)) || (rc = pl->d_pipe_lane_pair.upload(pl->pipe.lane_pair, st)) ||
            (rc = pl->d_pipe_bnd.alloc((size_t)pl->pipe.bnd_elems))) {
            delete pl;
            return rc;
        }
        for (const PipeItem &pi : pl->pipe.items) pl->pipe_analytic_rows = std::max(pl->pipe_analytic_rows, pi.rsteps + 16);
        if (want_paths) {
            // (k_trace_recompute prefetches up to a block and a few rows beyond a sequence's last row)
            pl->pipe_analytic_rows += PRALINE_KEEP_BH + 16;
            // sequences one per task: the set's, for the lanes that hold a pair
            std::vector<int32_t> l1(pl->pipe.lane_pair.size(), -1);
            for (const PipeItem &pi : pl->pipe.items)
                for (int t = pi.task0; t < pi.task0 + pi.ntasks; ++t)
                    for (int q = 0; q < 32; ++q)
                        if (pl->pipe.lane_pair[(size_t)t * 32 + q] >= 0) l1[(size_t)t * 32 + q] = pl->pipe.set_one[(size_t)pi.set * 32 + q];
            if ((rc = pl->d_pipe_lane_one.upload(l1, st)) || (rc = pl->d_pipe_analytic4.alloc((size_t)pl->pipe_analytic_rows * 32))) {
                delete pl;
                return rc;
            }
            if (hipStreamSynchronize(st) != hipSuccess) { delete pl; return fail(PRALINE_ERR_DEVICE, "plan upload failed"); }   // (l1 goes out of scope)
        }
        if ((rc = pl->d_pipe_analytic.alloc((size_t)pl->pipe_analytic_rows * 32))) { delete pl; return rc; }
        // (rows the kernels never write only feed padding rows; keep them free of NaN bit patterns)
        if (hipMemsetAsync(pl->d_pipe_bnd.p, 0, (size_t)pl->pipe.bnd_elems * sizeof(float2), st) != hipSuccess) {
            delete pl;
            return fail(PRALINE_ERR_DEVICE, "plan upload: memset failed");
        }
    }
    if (want_paths) {
        std::vector<int32_t> pv(pairs, pairs + 2 * n_pairs);
        if ((rc = pl->d_pairs.upload(pv, st)) || (rc = pl->d_loc.upload(loc, st)) ||
            (rc = pl->d_end_cells.alloc((size_t)n_pairs * 4)) || (rc = pl->d_path_rows.alloc((size_t)n_pairs)) ||
            (rc = pl->d_path_start.alloc((size_t)n_pairs)) || (rc = pl->d_paths.alloc((size_t)cap * 2)) ||
            (rc = pl->d_slot_off.upload(pl->slot_off, st))) {
            delete pl;
            return rc;
        }
        if (pl->has_rects) {
            std::vector<int32_t> ro(rect_off, rect_off + n_pairs + 1), rv(rects, rects + (size_t)rect_off[n_pairs] * 4);
            if ((rc = pl->d_rect_off.upload(ro, st)) || (rc = pl->d_rects.upload(rv, st))) { delete pl; return rc; }
        }
        if (pl->mask_kind == 2) {
            std::vector<int64_t> zo((size_t)n_pairs);
            int64_t tot = 0;
            for (int64_t p = 0; p < n_pairs; ++p) {
                zo[(size_t)p] = tot;
                tot += (int64_t)((a.len[pairs[2 * p + 1]] + 31) / 32) * (a.len[pairs[2 * p]] + 1);
            }
            if ((rc = pl->d_zm_off.upload(zo, st)) || (rc = pl->d_zmask.alloc((size_t)tot))) { delete pl; return rc; }
            hipLaunchKernelGGL(k_build_zmask, dim3((unsigned)n_pairs), dim3(256), 0, st, pl->d_pairs.p, a.d_len.p, pl->d_rect_off.p,
                               pl->d_rects.p, pl->d_zm_off.p, pl->d_zmask.p);
        }
    }
    pt.mark("allocations + uploads (async)");
    hipError_t e = hipStreamSynchronize(st);
    pt.mark("stream sync");
    if (e == hipSuccess) e = hipEventCreate(&pl->ev0);
    if (e == hipSuccess) e = hipEventCreate(&pl->ev1);
    if (e != hipSuccess) { delete pl; return fail(PRALINE_ERR_DEVICE, "plan upload: %s", hipGetErrorString(e)); }
    *out = pl;
    return PRALINE_OK;
}

extern "C" int praline_plan_destroy(praline_plan *plan)
{
    if (!plan) return PRALINE_OK;
    // (the device blocks go back to the stream-ordered pool; the explicit waits keep the plan's host-side state from
    // outliving work that still reads it)
    PhaseTimer pt("plan_destroy");
    if (g_rt.ready) { (void)hipStreamSynchronize(g_rt.stream); (void)hipStreamSynchronize(g_rt.stream2); }
    pt.mark("wait for both streams");
    delete plan;
    pt.mark("release");
    return PRALINE_OK;
}

extern "C" int64_t praline_plan_cells(const praline_plan *plan) { return plan ? plan->cells : 0; }
extern "C" int64_t praline_plan_steps(const praline_plan *plan)
{
    if (!plan) return 0;
    if (plan->pipe.ok && !plan->want_paths) return plan->pipe.steps;   // wave steps of the pipeline launch (idle waves of the last rounds included)
    int64_t steps = 0;
    for (const WaveTask &wt : plan->tasks)
        if (wt.max_l1 > 0) steps += (int64_t)wt.nstrips * (wt.max_l1 + 1);
    return steps;
}
extern "C" int64_t praline_plan_tasks(const praline_plan *plan)
{
    if (!plan) return 0;
    if (plan->pipe.ok && !plan->want_paths) return (int64_t)plan->pipe.tasks.size();
    int64_t n = 0;
    for (const WaveTask &wt : plan->tasks) n += wt.max_l1 > 0;
    return n;
}
extern "C" int64_t praline_plan_path_capacity(const praline_plan *plan) { return plan ? plan->path_cap : 0; }
extern "C" void *praline_plan_device_scores(praline_plan *plan) { return plan ? (void *)plan->d_scores.p : nullptr; }
