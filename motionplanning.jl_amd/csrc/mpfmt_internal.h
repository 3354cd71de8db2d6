// Internal declarations shared by the translation units of libmpfmt.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <map>
#include <vector>
#include "../../include/mpfmt.h"
#include "mpfmt_host.h"

#define MPFMT_WAVE 64           // CDNA wavefront width; tile of samples = one wavefront of queries
#define MPFMT_MAXS 16           // max candidate slices per tile

struct mpfmt_timer {
    double total_ms = 0.0;
    int64_t launches = 0;
};

// Geometry of the uniform cell grid that bins the samples for one radius (host copy; passed by value
// to kernels).  Cells are at least r wide in every gridded dimension, so the neighbours of a point in
// cell c lie in cells c-1..c+1.
// Cell ids.  Unsharded: row-major with the LAST dimension fastest, so a run of cells along that dimension is a contiguous
// range of the cell-sorted sample array.  Sharded (world > 1): BLOCK-MAJOR -- the leading nsplit axes are cut in two at a cell
// boundary (cells [0, split) | [split, g)), the id's most significant digits say which half of each cut axis a cell lies in
// (axis 0 first), and inside such a half-block the order is the row-major one above.  An index range of the cell-sorted order
// is then a compact block of space (0.5^3 x 1^3 at 8 shards in R^6) instead of a slab thinner than r, which is what decides
// how many edges join two shards.  The last axis is never cut (runs along it stay contiguous); a cut axis with an odd cell
// count leaves holes in the id range (ids of cells that do not exist: empty).
struct mpfmt_grid {
    int32_t gd;                          // dims are all gridded; g[i] == 1 means "not split"
    int32_t g[MPFMT_MAX_DIM];            // cells per dimension
    double  lo[MPFMT_MAX_DIM];           // lower corner of the sample bounding box
    double  w[MPFMT_MAX_DIM];            // cell width
    double  inv_w[MPFMT_MAX_DIM];
    int64_t stride[MPFMT_MAX_DIM];       // id stride per dimension inside a half-block
    int32_t split[MPFMT_MAX_DIM];        // 0: the axis is not cut; else the first cell of its upper half
    int32_t ext[MPFMT_MAX_DIM];          // cells per dimension of a half-block's id range (split, or g when not cut)
    int64_t hstride[MPFMT_MAX_DIM];      // id offset of the upper half of a cut axis
    int32_t nsplit;                      // cut axes (the leading ones: 0 .. nsplit-1)
    int64_t inner;                       // ids per half-block
    int64_t ncells;                      // id range = inner << nsplit
};
// contribution of cell coordinate c of axis i to the cell id
__host__ __device__ __forceinline__ int64_t mpfmt_cell_term(const mpfmt_grid& G, int i, int c)
{
    const int s = G.split[i];
    return (s > 0 && c >= s) ? G.hstride[i] + (int64_t)(c - s) * G.stride[i] : (int64_t)c * G.stride[i];
}
// cell coordinate of axis i out of a cell id (may be >= g[i] for a hole of the block-major id range)
__host__ __device__ __forceinline__ int mpfmt_cell_coord(const mpfmt_grid& G, int i, int64_t id)
{
    const int64_t in = id % G.inner;
    int c = (int)((in / G.stride[i]) % G.ext[i]);
    if (G.split[i] > 0 && ((id / G.hstride[i]) & 1)) c += G.split[i];
    return c;
}

struct mpfmt_boxes_dev {                 // obstacle set in HBM: [M][2][dw] (lo then hi per box)
    const double* lohi;
    int32_t M;
    int32_t dw;
};

struct mpfmt_ss {                        // BoundedStateSpace bounds (statespaces.jl:29-34)
    int32_t has;
    int32_t d;
    double lo[MPFMT_MAX_DIM];
    double hi[MPFMT_MAX_DIM];
};

// 2-D SAT world (kernels_sat2d.hip): a Circle or a convex Polygon with the fields the predicates read
#define MPFMT_MAX_POLY 16
struct mpfmt_shape2d {
    int32_t kind, n;
    double c[2], r;
    double xr[2], yr[2];
    double pts[MPFMT_MAX_POLY][2], normals[MPFMT_MAX_POLY][2], nex[MPFMT_MAX_POLY][2];
};
struct mpfmt_aabb2d { double xr[2], yr[2]; };

struct mpfmt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    // A second stream for the two kernels of a step that have nothing to wait for on the first but one predecessor: the per-sample obstacle
    // masks (beside the chunk lists) and the flagged pairs' exact tests (beside the logs' degree count and its scans).  Forked and joined
    // with events; side_pending: work is in flight there that ctx->stream has not waited for yet.
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool side_pending = false;
    int64_t preset_entries = -1;         // speculative step: the mask's trusted capacity, for the preset beside the exact pair tests
    int64_t mask_preset_words = -1;      // words of graph_free preset to ones ahead of mpfmt_order_logs (-1: none)
    bool masks_early = false;            // this build's sample masks were launched beside its chunk lists
    int32_t overlap = 1;                 // option: 1 = the side stream from 65536 samples on, 2 = always, 0 = every kernel of the step on ctx->stream
    hipStream_t copy_stream[2] = {nullptr, nullptr};      // mpfmt_graph_export: two device-to-host streams and their hand-over events
    hipEvent_t ev_conv[2] = {nullptr, nullptr}, ev_copy[2] = {nullptr, nullptr};
    void* export_arena = nullptr;        // page-locked host memory of mpfmt_graph_export_pinned (grow-only, lives as long as the ctx)
    size_t export_arena_bytes = 0;
    std::string err;
    int rank = 0, world = 1;

    // ---- samples -------------------------------------------------------------------------------
    int64_t N = 0;
    int32_t d = 0;
    double* Xo = nullptr;                // [N][d] original order (AoS = the caller's layout)
    double* Xo_next = nullptr;           // where an upload lands: changes places with Xo once the set has been accepted (a refused set leaves the ctx as it was)
    double bb_lo[MPFMT_MAX_DIM], bb_hi[MPFMT_MAX_DIM];
    void* bb_dev = nullptr;              // mpfmt_upload_samples_device: per-block partial boxes, and their pinned host mirror
    void* bb_host = nullptr;

    // ---- cell grid for radius grid_r --------------------------------------------------------------
    double grid_r = -1.0;
    mpfmt_grid grid;
    int64_t ntiles = 0;                  // ceil(N/64)
    int32_t* perm = nullptr;             // [ntiles*64] sorted position -> original index (pad = -1)
    int32_t* iperm = nullptr;            // [N] original index -> sorted position
    uint32_t* cellkey = nullptr;         // [N] cell id of each sorted position
    // one arena, one fill per index build: cellstart [ncells + 1] (the cells' counters, scanned in place) | list_max (longest chunk list,
    // k_chunk_lists) | tileneed [ntiles] bytes
    void* idx_arena = nullptr;
    int32_t* cellstart = nullptr;        // [ncells+1] (inside idx_arena)
    int32_t* list_max = nullptr;         // (inside idx_arena)
    int32_t* cellcnt_pad = nullptr;      // block-major ids (sharded ctx): the cell counters of the count pass, one per 64-byte line (kernels_rdisc.hip)
    bool list_max_clean = false;         // zeroed by the index build's fill and not written since
    // sharded ctx on the matrix-core path: only the tiles this rank reads are built -- its own and the halo (the tiles of the cells next
    // to its own cells); tileneed [ntiles] marks them (nullptr: the index is whole)
    uint8_t* tileneed = nullptr;
    uint8_t* tileneed_buf = nullptr;     // (inside idx_arena)
    int32_t index_halo = 1;              // option: allow the shard + halo index
    int32_t shard_blocks = 1;            // option: block-major cell ids on a sharded ctx (0: row-major -- shards are slabs; measurements)
    int index_rank = 0, index_world = 1; // the shard the index was built for
    std::vector<double> cut_frac;        // shard boundaries as fractions of the cell-sorted order (cut_key: the geometry they belong to)
    std::vector<int64_t> cut_key;
    double* Xt = nullptr;                // [ntiles][d][64] tiled SoA, cell-sorted, NaN padded
    double* tile_lo = nullptr;           // [ntiles][d] tight bounding box of each tile
    double* tile_hi = nullptr;
    double* tile_sub = nullptr;          // [ntiles][4][d] two sub-boxes per tile (k_tile_bbox)
    float* tile_sub32 = nullptr;         // the same boxes in fp32, rounded outward: the candidate side of the chunk-list test (half the bytes)

    // ---- r-disc graph (device resident) ------------------------------------------------------------
    double graph_r = -1.0;
    bool graph_counted = false, graph_filled = false;
    int32_t S = 1;                       // candidate slices per tile
    int64_t tile_begin = 0, tile_end = 0;  // shard tile range
    int32_t* slice_cnt = nullptr;        // [S][ntiles*64] hits per (slice, sorted query)
    int64_t* deg = nullptr;              // [N+1] degree by original index
    int64_t* degs = nullptr;             // [npad+1] degree by sorted position
    int64_t* tptr = nullptr;             // [npad+1] offsets of the sorted-order staging CSC (rowtmp/valtmp)
    bool tptr_valid = false;             // tptr holds the scan of the counted graph's degs (the single-pass build leaves it undone)
    // MFMA filter path (kernels_rdisc_mfma.hip)
    double* Xs = nullptr;                // [npad][d] cell-sorted AoS fp64 (NaN padded): exact refine gathers
    void* ops = nullptr;                 // [npad] 16 fp16 slots (32 B) per sorted sample: MFMA operands
    double mf_scale = 1.0;
    double ops_r = -1.0;                 // grid radius the operands were built for
    int32_t rdisc_path = 0;              // 0 auto, 1 exact fp64 VALU kernel, 2 MFMA filter + exact refine
    int32_t rdisc_path_used = 0;
    bool filter_valu = false;            // the counted graph's pair kernel ran the exact fp64 filter on the vector ALUs (k_rdisc_vf_w4) instead of the fp16 matrix-core one
    int32_t lists_wide = -1;             // (fixed; was an option until the A/B was settled) chunk lists built by four wavefronts per tile on small shards, one elsewhere
    int32_t cell_fb_max = 8;             // position bits inside a cell that the sort key carries (k_cellkey)
    int64_t mf_tail_min_items = 32768;   // ... in launches of at least this many items (smaller ones do not fill the chip: nothing to even out)
    int32_t* ord_ctr = nullptr;          // [8] inside the counter arena: the ordering kernel's per-XCD quarter counters (zeroed with the arena)
    int32_t ord_draw = 1;                // option: 1 = the ordering kernel's workgroups draw their quarters when those are long (>= 1536 records on average), 2 = always, 0 = every nb-th quarter each
    int32_t mf_tail_permille = 80, mf_tail_slices = 9;      // options: the last tiles of a single-pass pair-kernel launch are cut into this many slices (0: off)
    int32_t mf_xcd_mode = -1;            // work items go to the XCDs in interleaved groups of this many; -1: 256 for launches of >= 32768 items, else 64
                                         // (north star: groups of 64 2.02 ms / 5.6 GB of counter traffic, 256 2.04 / 4.6, 512 2.05 / 4.4; one range per XCD 2.41 ms)
    int32_t num_cus = 256;               // compute units of the device (persistent-grid sizing)
    int ord_per_cu = 0;          // k_order_logs: resident workgroups per CU on THIS ctx's device
    int* sweep_ctr = nullptr;            // graph sweep: one task counter per obstacle chunk
    int32_t sweep_rounds = 1;            // option: round-table sweep (k_graph_sweep_rt) where it applies (d <= 8, M <= 256)
    int64_t* rt_cnt = nullptr;           // [columns visited + 1] rounds per column, then (scan) first round of each column
    int64_t* rt_off = nullptr;
    void* rt_tmp = nullptr;              // scan temporary
    void* rt_table = nullptr;            // [rounds] (column, entries | first << 31, first entry) in visiting order
    int64_t* rt_total = nullptr;         // device: number of rounds
    double* rt_ss = nullptr;             // device copy of the state-space bounds (lo[MAX_DIM], hi[MAX_DIM]) for scalar loads
    mpfmt_ss rt_ss_host;                 // what rt_ss holds
    bool rt_ss_valid = false;
    // "every sample lies in the state space" for (samples_epoch, ss): the sweep then skips the per-row in_state_space test
    int64_t samples_epoch = 0, ssflag_epoch = -1;
    mpfmt_ss ssflag_ss;
    bool ssflag_all_in = false;
    int32_t* ssflag_dev = nullptr;
    int64_t mf_target_items = 40000;     // work items (tile x slice) the MFMA path aims for (tools/run_shard_sweep_items.py: flat from 40k up at 1 shard, best at 2 and 4)
    float mf_negT = 0.f;
    void* lists = nullptr;               // [shard tiles][list_cap] candidate chunk ids per tile
    int32_t* list_len = nullptr;         // [shard tiles + 1] lengths, last = max
    void* lists_stage = nullptr;         // small shards with long lists: [tiles][4][list_cap] staging of the 4-wavefront list kernel
    int64_t list_cap = 0;
    double lists_r = -1.0; int64_t lists_begin = -1, lists_end = -1;
    // single-pass hit pool (MFMA path): hits found by the count pass are kept, so the fill pass is a scatter
    int32_t use_pool = 1;                // option "rdisc_pool"
    int32_t* pool_flag = nullptr;        // overflow flag
    int64_t qcap = 0;                    // capacity of one quarter log, in records (a multiple of 16)
    uint32_t* qkey = nullptr;            // [quarter tiles of the shard][qcap] record keys: row sample index | column within the quarter << 26 | flags
    double* qd2 = nullptr;               // [quarter tiles of the shard][qcap] squared distances
    int32_t* qlen = nullptr;             // [quarter tiles of the shard] the logs' cursors
    const double* st_C = nullptr; const uint64_t* st_H = nullptr; int st_free = 0;      // streaming mode (mpfmt_rdisc_stream): inputs of the launch in flight
    void* st_best = nullptr; int32_t* st_besti = nullptr; int32_t* st_nfree = nullptr;  //   and its per-slice partials [S][npad]
    void* smask = nullptr;               // [npad] per-sample obstacle masks (k_sample_masks): the drain's broad phase walks the boxes in (mask_q & mask_c) only
    int64_t pool_hint_qmax = 0;          // records in the fullest quarter (16 consecutive cell-sorted columns) of the last build: sizes the next one's logs
    // half build of the single-pass r-disc graph (kernels_rdisc_mfma.hip: every pair found once, the other column's record goes
    // to a foreign log of that column's tile)
    int use_half = 1;                    // option rdisc_half
    int fuse_broad = 2;                  // option: broad phase of the edge tests in the half build's drain (step APIs only)
    void* pend_items = nullptr;          // [segments][pend_wcap] (entry, column sample, row position) of the entries that need an exact test
    int64_t pend_wcap = 0;
    int32_t* pend_cnt = nullptr;         // [segments] items per segment, then the overflow flag
    int32_t* pend_over = nullptr;
    int pend_nseg = 0;
    bool sweep_pending_used = false;     // the last graph sweep visited the pending list only
    bool pend_overflowed = false;        // read back behind the speculative step's synchronisation
    bool pend_valid = false;             // the resident graph has its pending list (made by this step's ordering pass)
    void* pair_items = nullptr;          // fuse_broad = 2: [items][pair_icap] 32-byte pending-pair items (k_exact_pairs)
    int32_t* pair_cnt = nullptr;         // [items], then the overflow flag
    int32_t* pair_over = nullptr;
    int64_t pair_icap = 0;
    int pair_slack = 1;                  // doubled (to 8) by a step whose pending-pair list overflowed
    bool sweep_in_order = false;         // the mask of the resident graph was written by the ordering pass (form 2)
    bool bits_in_records = false;        // this count's blocked edges are marked in the records (bit 31 of the row index)
    int debug_small_lists = 0;           // option (tests): pending lists of 8 items, so that their overflow path runs
    bool want_broad = false;             // set by the step APIs around their count
    bool broad_in_drain = false;         // this count's records carry the broad-phase flag (bit 30 of the row index)
    bool half_used = false;              // the counted graph was built that way
    int64_t redo_count = 0; int32_t redo_reason = 0;      // stats: builds redone because a capacity did not hold (1 chunk list cut, 2 log overflow, 4 column too long, 8 nnz beyond the allocation, 16 pending list cut)
    bool pool_skip_once = false;         // the next count runs in the two-pass form (a log of the single pass overflowed, or a column was too long)
    bool lists_half = false;             // the cached chunk lists hold only chunks >= the tile
    int cell_fb = 0;                     // position bits below the cell id in cellkey (k_cellkey)
    int64_t max_deg = 0;                 // longest column of the counted graph (k_degree)
    int32_t pool_slack = 1;              // doubled after a build whose slot lists overflowed
    bool pool_valid = false;             // pool holds exactly the nnz hits of the counted graph
    // speculative single-sync step (mpfmt_graph_step): capacities of the previous identical build are trusted, every kernel
    // after the count bails out on the device flag spec_fail, and the host validates once at the end
    bool spec_ready = false;             // the previous build of the same (N, r, shard) went through the single-pass pool path
    bool spec_lists = false;             // chunk lists of the pending count were built without reading back their maximum
    bool cnt_pool = false, cnt_mf = false;   // the pending count used the pool / the MFMA pair kernel
    int32_t* spec_fail = nullptr;        // device flag: pool overflow, truncated chunk list or nnz beyond the trusted capacity
    int64_t nnz_cap = 0;                 // entries rowval / nzval / the mask are sized for
    void* rb_dev = nullptr;              // count read-back block (device) and its pinned host mirror
    void* rb_host = nullptr;
    int64_t lists_cap_trusted = -1;      // list capacity that a verified build found sufficient
    int64_t pool_hint_N = -1; double pool_hint_r = -1.0; int64_t pool_hint_nnz = 0; int64_t pool_hint_maxdeg = 0; int pool_hint_rank = -1, pool_hint_world = -1;   // capacity hint from the last build
    int64_t survivors = 0;
    int64_t* colptr = nullptr;           // [N+1] 0-based offsets by original index
    int64_t nnz = 0;
    int32_t* rowtmp = nullptr;           // [nnz] unsorted fill
    double* valtmp = nullptr;
    int32_t* rowval = nullptr;           // [nnz] 0-based, ascending per column
    double* nzval = nullptr;
    int32_t* rowpos = nullptr;           // [nnz] cell-sorted position of each entry's row (single-pass build): the sweep gathers rows from Xs
    bool rowpos_valid = false;
    int32_t sweep_sorted = 1;            // (fixed; was an option until the A/B was settled) gather the sweep's rows from Xs in cell-sorted order with per-XCD task ranges (6x less HBM traffic,
                                         // 74 % L2 hits): with the round-table sweep, whose instruction count no longer hides under the
                                         // caller-order gather (2.29 ms floor), this is the faster mode (2.2 vs 2.65 ms); on by default
    uint64_t* graph_free = nullptr;      // [ceil(nnz/64)]
    bool graph_swept = false;
    size_t zarena_bytes = 0;
    bool deg_zero_valid = false;         // sharded ctx: deg[] is all zeros (the ordering pass cleared what the last step wrote)
    void* zarena = nullptr;              // one arena for d_pairs, pool_flag, pair_cnt (one fill per build); they point into it when it exists
    unsigned long long* d_pairs = nullptr;   // device counter: candidate pairs tested
    int64_t pairs_tested = 0;

    // ---- double-integrator (LinearQuadratic) graph: shares colptr / rowval / nzval / graph_free -------------
    int32_t di_S = 1;
    double di_rho = 1.0, di_r = 0.0;
    bool di_counted = false, di_filled = false, di_swept = false;
    int32_t* di_pool_i = nullptr; double* di_pool_c = nullptr; double* di_pool_t = nullptr;   // DI single-pass slot lists
    int64_t di_pool_cap = 0; bool di_pool_valid = false;
    void* di_ops = nullptr;              // matrix-core prefilter of the double-integrator build (kernels_di_mfma.hip): target- and source-role operands
    bool di_mf = false; float di_negT = 0.f;     // the counted DI graph went through it; its threshold
    int32_t di_path = 0;                 // option: 0 auto, 1 vector-ALU candidate test, 2 matrix-core prefilter
    int32_t steer_kind = 1;              // which steering graph the di_* state describes: 1 double integrator, 2 Dubins car, 3 Reeds-Shepp car
    double car_rt = 1.0, car_sp = 1.0;   // Dubins turning radius / speed of the built graph
    uint64_t* car_keep = nullptr;        // keep bits over the candidate (positions) graph
    mpfmt_ctx* aux = nullptr;            // helper ctx: Euclidean r-disc graph of the positions (Dubins build)
    double* tvaltmp = nullptr;           // [nnz] optimal times, unsorted staging
    double* tval = nullptr;              // [nnz] optimal times t* per entry
    uint8_t* di_nseg = nullptr;          // [nnz] workspace segment tests the reference would have made per edge

    // ---- obstacles -----------------------------------------------------------------------------
    double* boxes = nullptr;             // [M][2][dw]
    std::vector<double> boxes_host;      // the same on the host
    int32_t M = 0, dw = 0;
    bool have_boxes = false;
    int32_t cc_kind = 0;                 // collision checker: 0 = PointRobotNDBoxes, 1 = PointRobot2D (SAT)
    mpfmt_shape2d* shapes2d = nullptr;   // [M] when cc_kind == 1
    mpfmt_aabb2d aabb2d;                 // Compound2D bounding box
    mpfmt_ss ss;

    // ---- scratch -------------------------------------------------------------------------------
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    std::map<void*, size_t> caps;       // capacity (bytes) of each grow-only device buffer, keyed by member address
    std::map<std::string, mpfmt_timer> timers;
    void* timer_state = nullptr;         // HIP-event timing records of this ctx (mpfmt_capi.hip)
    bool timing_enabled = true;
    bool rebuild_index = false;          // option "rebuild_index": graph_build_device rebuilds the cell grid every call

    // ---- multi-GPU exchange (mpfmt_comm.hip) and the device-resident wavefront FMT* driver (kernels_wavefront.hip) ----
    void* comm = nullptr;                // mpfmt_comm: RCCL communicator + communication stream of this ctx
    int32_t step_state = 0; double step_r = 0.0;   // graph_step_launch / _finish: 0 none, 1 speculative kernels in flight, 2 complete
    int32_t wf_graphs = 0;               // option (off: measured, no gain -- the solve is bound by k_wf_connect, 3.2 of 5.4 ms, not by launches): replay a captured
                                         // hipGraph of a group of 8 steps instead of launching its ~48 kernels
    int32_t wf_pos_space = 1;            // option: the device solve gathers its sets by cell-sorted position from the second solve on a graph
                                         // on (0: always by caller index, 2: from the first solve -- measurements, tests)
    int64_t wf_seen_epoch = -1; double wf_seen_r = -1.0; int64_t wf_seen_nnz = -1;      // the graph the last device solve ran on
    int32_t wf_pos_used = 0;             // stat: the last device solve gathered by position
    int32_t wf_force_sharded = 0;        // option: run the sharded form of the wavefront step (own-column marking, triples, exchange) at world = 1
    void* wf = nullptr;                  // mpfmt_wf: W / H / C / A and the batch lists of a running wavefront solve
};

// error helpers ---------------------------------------------------------------------------------
int32_t mpfmt_fail(mpfmt_ctx* ctx, int32_t code, const char* fmt, ...);
#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return mpfmt_fail((ctx), MPFMT_ERR_HIP, "%s failed: %s (%s:%d)", #call,                \
                              hipGetErrorString(e_), __FILE__, __LINE__);                          \
    } while (0)

int32_t mpfmt_scratch(mpfmt_ctx* ctx, size_t bytes, void** out);
// grow-only device buffer: (re)allocates *p when it is smaller than bytes
int32_t mpfmt_ensure(mpfmt_ctx* ctx, void** p, size_t bytes);
void mpfmt_time_begin(mpfmt_ctx* ctx);
void mpfmt_time_end(mpfmt_ctx* ctx, const char* name);
void mpfmt_time_abandon(mpfmt_ctx* ctx);
// one timed interval; an error return between begin and end abandons it (the stack depth and the opening event are given
// back), so failed calls cannot wedge the timers of the ctx
struct mpfmt_timed {
    mpfmt_ctx* ctx; bool open;
    explicit mpfmt_timed(mpfmt_ctx* c) : ctx(c), open(true) { mpfmt_time_begin(c); }
    void end(const char* name) { if (open) { mpfmt_time_end(ctx, name); open = false; } }
    ~mpfmt_timed() { if (open) mpfmt_time_abandon(ctx); }
    mpfmt_timed(const mpfmt_timed&) = delete;
    mpfmt_timed& operator=(const mpfmt_timed&) = delete;
};

// mpfmt_capi.hip: the side stream (see mpfmt_ctx::side_stream)
int32_t mpfmt_side_fork(mpfmt_ctx* ctx, hipStream_t* main_out);      // ctx->stream := the side stream, ordered after everything issued so far
int32_t mpfmt_side_back(mpfmt_ctx* ctx, hipStream_t main);           // ctx->stream := main again; the side work is pending
int32_t mpfmt_side_join(mpfmt_ctx* ctx);                             // ctx->stream waits for the pending side work (no-op without)

// kernels_rdisc.hip -----------------------------------------------------------------------------
int32_t mpfmt_build_grid(mpfmt_ctx* ctx, double r, bool whole = false);      // whole: every tile is built even on a sharded ctx
int32_t mpfmt_launch_rdisc_count(mpfmt_ctx* ctx, double r);
int32_t mpfmt_rdisc_count_launch(mpfmt_ctx* ctx, double r, bool spec);
int32_t mpfmt_rdisc_count_finish(mpfmt_ctx* ctx, double r, bool* spec_failed);
int32_t mpfmt_graph_step(mpfmt_ctx* ctx, double r);
int32_t mpfmt_graph_step_launch_impl(mpfmt_ctx* ctx, double r);
int32_t mpfmt_graph_step_finish_impl(mpfmt_ctx* ctx);
int32_t mpfmt_launch_rdisc_fill(mpfmt_ctx* ctx, double r);
int32_t mpfmt_mfma_prepare(mpfmt_ctx* ctx, double r, float* negT_out, bool* usable);
int32_t mpfmt_mfma_build_operands(mpfmt_ctx* ctx);
template <int MODE> int32_t mpfmt_launch_rdisc_mfma(mpfmt_ctx* ctx, double r, float negT);
int32_t mpfmt_order_logs(mpfmt_ctx* ctx, const int32_t* spec_fail = nullptr, int64_t mask_entries = -1);      // kernels_order.hip
int32_t mpfmt_mask_preset(mpfmt_ctx* ctx, int64_t entries);
int32_t mpfmt_sweep_prepare_ss(mpfmt_ctx* ctx);          // kernels_sweep.hip: device copy of the state-space bounds + the all-samples-inside flag
#define MPFMT_ORD_MAXDEG 2048        // longest column the log-ordering kernel stages in LDS (ORD_STG in kernels_order.hip)
int32_t mpfmt_mfma_build_lists(mpfmt_ctx* ctx, double r, bool* usable, bool spec = false, bool half = false);
int mpfmt_slices_for(const mpfmt_ctx* ctx, int64_t units, bool mfma);      // slices per tile of the pair kernels (odd: kernels_rdisc_mfma.hip)
int32_t mpfmt_launch_log_degrees(mpfmt_ctx* ctx);
int32_t mpfmt_launch_sample_masks(mpfmt_ctx* ctx, double r, void* zero = nullptr, size_t zero_bytes = 0, bool* zeroed = nullptr);      // (zero: a buffer the kernel clears on the side)
int32_t mpfmt_rdisc_stream_impl(mpfmt_ctx* ctx, double r, const double* C_host, const uint64_t* H_host, int32_t want_free,
                                int64_t* deg, int64_t* nfree, int64_t* parent, double* cost, int64_t* nnz_out);
int32_t mpfmt_launch_exact_pairs(mpfmt_ctx* ctx, const int32_t* spec_fail);
// the library's own exclusive scan of int64 items on ctx->stream (k_scan_block / _single / _add, kernels_rdisc.hip); in == out allowed
size_t mpfmt_scan_tmp_bytes(size_t n);
int32_t mpfmt_scan_i64_tmp(mpfmt_ctx* ctx, const int64_t* in, int64_t* out, size_t n, void* tmp);      // tmp: mpfmt_scan_tmp_bytes(n) bytes of device memory
int32_t mpfmt_scan_i64(mpfmt_ctx* ctx, const int64_t* in, int64_t* out, size_t n);                     // tmp from the ctx's scratch buffer
int32_t mpfmt_launch_rdisc_query(mpfmt_ctx* ctx, int64_t v0, double r, int64_t* k_out,
                                 int64_t* inds_host, double* ds_host, int64_t cap);

// kernels_sweep.hip -----------------------------------------------------------------------------
int32_t mpfmt_launch_points_free(mpfmt_ctx* ctx, const int64_t* d_idx1, int64_t n, uint64_t* d_mask);
int32_t mpfmt_2d_launch_points(mpfmt_ctx* ctx, const double* X, const int64_t* idx1, int64_t n, uint64_t* d_mask);
int32_t mpfmt_2d_launch_edges(mpfmt_ctx* ctx, const int64_t* s1, const int64_t* t1, const double* P, const double* Q, int64_t E, uint64_t* d_mask);
int32_t mpfmt_2d_launch_graph(mpfmt_ctx* ctx);
int32_t mpfmt_launch_states_free(mpfmt_ctx* ctx, const double* d_P, int64_t n, uint64_t* d_mask);
int32_t mpfmt_launch_edges_free(mpfmt_ctx* ctx, const int64_t* d_src1, const int64_t* d_dst1, int64_t E, uint64_t* d_mask);
int32_t mpfmt_launch_motions_free(mpfmt_ctx* ctx, const double* d_P, const double* d_Q, int64_t n, uint64_t* d_mask);
int32_t mpfmt_launch_mc_edges(mpfmt_ctx* ctx, const int64_t* d_src1, const int64_t* d_dst1, int64_t E, double sigma, int64_t rollouts,
                              uint64_t seed, unsigned long long* d_hits);
int32_t mpfmt_launch_mc_is_edges(mpfmt_ctx* ctx, const int64_t* d_src1, const int64_t* d_dst1, int64_t E, double sigma, int64_t rollouts,
                                 uint64_t seed, unsigned long long* d_wsum);
int32_t mpfmt_launch_mc_ais_edges(mpfmt_ctx* ctx, const int64_t* d_src1, const int64_t* d_dst1, int64_t E, double sigma, int64_t rollouts,
                                  uint64_t seed, unsigned long long* d_wsum, double* d_mu);
int32_t mpfmt_launch_graph_sweep(mpfmt_ctx* ctx, const int32_t* spec_fail = nullptr, int64_t mask_entries = -1);

// kernels_di.hip ----------------------------------------------------------------------------------
#include <functional>
#include <vector>
int32_t mpfmt_csc_transpose_device(mpfmt_ctx* ctx, mpfmt_csr_host* out);      // kernels_di.hip; needs nnz < 2^32
int32_t mpfmt_csc_transpose_resident(mpfmt_ctx* ctx, int64_t* d_rowptr, int32_t* d_colidx);   // the same, result left on the device
int32_t mpfmt_car_build(mpfmt_ctx* ctx, int kind, double rt, double sp, double r);      // kind 1 Dubins, 2 Reeds-Shepp
int32_t mpfmt_car_sweep(mpfmt_ctx* ctx);
int32_t mpfmt_car_steer_batch(mpfmt_ctx* ctx, int kind, const double* d_X0, const double* d_X1, int64_t n, double rt, double sp, double* d_cost,
                              double* d_ctrl, int32_t* d_nseg);
struct di_args;
int32_t mpfmt_di_mf_prepare(mpfmt_ctx* ctx, double rho, double r, float* negT, bool* usable, double* sp_out, double* sv_out, double* pc);      // kernels_di_mfma.hip
int32_t mpfmt_di_mf_build_operands(mpfmt_ctx* ctx, double sp, double sv, const double* pc_host);
int32_t mpfmt_di_mf_launch(mpfmt_ctx* ctx, const di_args& a, int mode, float negT, unsigned nblk);
int32_t mpfmt_di_count(mpfmt_ctx* ctx, double rho, double r);
int32_t mpfmt_di_fill(mpfmt_ctx* ctx);
int32_t mpfmt_di_sweep(mpfmt_ctx* ctx);
int32_t mpfmt_di_steer_launch(mpfmt_ctx* ctx, int m, const double* dX0, const double* dX1, int64_t n, double rho, double r,
                              double* dcost, double* dt);

// mpfmt_comm.hip -----------------------------------------------------------------------------------
int32_t mpfmt_comm_allgather_inplace(mpfmt_ctx* ctx, void* buf, size_t bytes_per_rank, hipStream_t stream);
int32_t mpfmt_comm_world(const mpfmt_ctx* ctx, int* rank, int* world);      // 1 when a communicator exists

// kernels_steer.hip ---------------------------------------------------------------------------------
int32_t mpfmt_launch_euclid_steer(mpfmt_ctx* ctx, const int64_t* d_src1, const int64_t* d_dst1, int64_t E, double* d_t, double* d_u);
int32_t mpfmt_launch_euclid_propagate(mpfmt_ctx* ctx, const int64_t* d_src1, int64_t E, const double* d_t, const double* d_u,
                                      const double* d_s, double* d_out);

// kernels_wavefront.hip -----------------------------------------------------------------------------
void mpfmt_wf_free(mpfmt_ctx* ctx);
int32_t mpfmt_wf_run(mpfmt_ctx* ctx);
int32_t mpfmt_wf_begin_directed(mpfmt_ctx* ctx, int64_t init_idx, int32_t checkpts, const uint64_t* F_host, int32_t goal_kind,
                                const double* goal_params, int32_t gd, double band, int32_t flags);
void mpfmt_wf_info_now(mpfmt_ctx* ctx, mpfmt_wf_info* info);

// kernels_expand.hip ----------------------------------------------------------------------------
int32_t mpfmt_launch_expand(mpfmt_ctx* ctx, const uint64_t* d_W, const uint64_t* d_H, const uint64_t* d_F,
                            const double* d_C, const int64_t* d_zs1, int64_t nz,
                            int64_t* d_xs, int64_t* d_ymin, double* d_cmin, uint8_t* d_free, int64_t cap,
                            int64_t* nx_host);
