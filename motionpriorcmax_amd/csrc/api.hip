// libmpcmax: shape validation, workspace layout, version and error reporting.
#include "common.h"
#include "bounds.h"
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void mpc_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int mpc_version(void) { return MPC_VERSION; }

// ---- diagnostics: per-kernel HIP-event timer (common.h: MPC_LAUNCH) -----------------------------------------------
// Process-wide on purpose (the backward of a torch.autograd.Function runs on another thread than its forward); guarded by
// a mutex; off unless mpc_profile_start() was called, and then every launch costs two hipEventRecord calls.
#include <mutex>
#include <vector>
struct mpc_prof_rec { const char *name; hipEvent_t a, b; };
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;
static std::vector<mpc_prof_rec> g_prof;
static thread_local hipEvent_t g_prof_pending = nullptr;
bool mpc_prof_on() { return g_prof_on.load(std::memory_order_relaxed); }
void mpc_prof_pre(hipStream_t st) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { g_prof_pending = nullptr; return; }
    (void)hipEventRecord(e, st);
    g_prof_pending = e;
}
void mpc_prof_post(const char *name, hipStream_t st) {
    hipEvent_t a = g_prof_pending, e = nullptr;
    g_prof_pending = nullptr;
    if (a == nullptr) return;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipEventDestroy(a); return; }
    (void)hipEventRecord(e, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back({name, a, e});
}
extern "C" int mpc_profile_start(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_on.store(true);
    return 0;
}
extern "C" int mpc_profile_stop(char *names, int32_t names_cap, float *ms, int32_t cap) {
    g_prof_on.store(false);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0, pos = 0;
    for (auto &r : g_prof) {
        float t = 0.f;
        (void)hipEventSynchronize(r.b);
        const bool ok = hipEventElapsedTime(&t, r.a, r.b) == hipSuccess;
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
        const int len = (int)strlen(r.name);
        if (ok && names && ms && n < cap && pos + len + 1 < names_cap) {
            memcpy(names + pos, r.name, len); names[pos + len] = '\n'; pos += len + 1;
            ms[n++] = t;
        }
    }
    if (names && names_cap > 0) names[pos < names_cap ? pos : names_cap - 1] = 0;
    g_prof.clear();
    return n;
}
extern "C" const char *mpc_last_error_string(void) { return g_err; }

// ---- -DMPC_BOUNDS: the bounds-checked debug build (bounds.h) -------------------------------------------------------
// Every translation unit with checked accessors registers a reader of its violation record; mpc_bounds_check() waits for the
// device, reads and clears them all and returns the number of violations (0: clean; -1: this is not a bounds build); the first
// violation's file:line, workgroup, index and extent go to mpc_last_error_string().
#ifdef MPC_BOUNDS
static mpc_bounds_unit *g_bounds_units = nullptr;
void mpc_bounds_register(mpc_bounds_unit *u) { u->next = g_bounds_units; g_bounds_units = u; }
#endif
extern "C" int mpc_bounds_check(void) {
#ifdef MPC_BOUNDS
    (void)hipDeviceSynchronize();
    int total = 0;
    bool said = false;
    for (mpc_bounds_unit *u = g_bounds_units; u; u = u->next) {
        int r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (u->read(r, 1)) { mpc_set_error("mpc_bounds_check: cannot read the record of %s", u->file); return -2; }
        if (r[0] > 0 && !said) {
            const long long idx = ((long long)r[4] << 32) | (unsigned)r[3], ext = ((long long)r[6] << 32) | (unsigned)r[5];
            mpc_set_error("MPC_BOUNDS: %d violation(s) in %s, first at line %d, workgroup %d: index %lld, extent %lld", r[0], u->file, r[1], r[2], idx, ext);
            said = true;
        }
        total += r[0];
    }
    return total;
#else
    return -1;
#endif
}

__global__ __launch_bounds__(256) void k_zero_words(unsigned *__restrict__ p, size_t n) {
    // up to 3 head words bring the pointer to 16-byte alignment, then 16-byte stores, then up to 3 tail words
    size_t head = ((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) >> 2;
    if (head > n) head = n;
    const size_t n4 = (n - head) >> 2;
    uint4 *p4 = reinterpret_cast<uint4 *>(p + head);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p4[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) p[threadIdx.x] = 0u;
        const size_t t0 = head + (n4 << 2);
        if (t0 + threadIdx.x < n) p[t0 + threadIdx.x] = 0u;
    }
}

int mpc_zero_async(void *ptr, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return 0;
    if ((bytes & 3) || (reinterpret_cast<uintptr_t>(ptr) & 3)) { mpc_set_error("mpc_zero_async: unaligned"); return MPC_E_SHAPE; }
    const size_t n = bytes >> 2;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    MPC_LAUNCH(k_zero_words, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned *)ptr, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mpc_set_error("mpc_zero_async: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

int mpc_validate_shape(const mpc_shape *s) {
    MPC_CHECK_ARG(s->B >= 0 && s->M >= 0 && s->Mp >= 0 && s->Mp <= s->M, MPC_E_SHAPE, "bad B/M/Mp");
    MPC_CHECK_ARG(s->nb >= 1 && s->T >= 1, MPC_E_SHAPE, "need num_bins >= 1 and num_tref >= 1");
    MPC_CHECK_ARG(s->H >= 3 && s->W >= 3, MPC_E_SHAPE, "image must be at least 3x3 (reflect padding)");
    MPC_CHECK_ARG(s->sp >= 1, MPC_E_SHAPE, "lut_superpixel_size must be >= 1");
    MPC_CHECK_ARG(s->hq == (s->H + s->sp - 1) / s->sp && s->wq == (s->W + s->sp - 1) / s->sp, MPC_E_SHAPE,
                  "hq/wq must equal ceil(H/sp), ceil(W/sp)");
    MPC_CHECK_ARG(s->n >= 0 && s->K >= 0, MPC_E_SHAPE, "bad n/K");
    MPC_CHECK_ARG((int64_t)s->B * s->nb * s->hq * s->wq * s->T < (1LL << 30), MPC_E_UNSUPPORTED, "LUT too large");
    MPC_CHECK_ARG((int64_t)s->B * s->M < (1LL << 40), MPC_E_UNSUPPORTED, "too many events");
    if (s->flags & (MPC_F_SCALE_BY_DT | MPC_F_POLARITY_SPLIT)) {
        // focus.py:49-50
        MPC_CHECK_ARG(s->T == 1, MPC_E_SHAPE, "scale_iwe_by_dt / polarity_aware_batching require num_tref == 1");
    }
    return 0;
}

mpc_ws_layout mpc_layout(const mpc_shape *s) {
    mpc_ws_layout L;
    memset(&L, 0, sizeof(L));
    L.P = (s->flags & MPC_F_POLARITY_SPLIT) ? 2 : 1;
    L.nimg = s->B * s->T * L.P;
    L.G = s->hq * s->wq;
    int64_t off = 0;
    // sized for the finest of the contrast tilings (marching kernel: 56-wide bands of MPC_CT_H / 2 rows)
    L.n_cblocks = mpc_cdiv(s->W, MPC_CF_TW) * mpc_cdiv(s->H, MPC_CT_H / 2) * (L.nimg > 0 ? L.nimg : 1);
    L.off_cpart = off; off += mpc_align((int64_t)L.n_cblocks * 2 * sizeof(double));
    L.n_sblocks_max = mpc_cdiv(s->wq, MPC_SM_W) * mpc_cdiv(s->hq, MPC_SM_H / 2) * (s->B > 0 ? s->B : 1) * s->nb * s->T;    // (bands of MPC_SM_H / 2 rows at most)
    L.off_spart = off; off += mpc_align((int64_t)L.n_sblocks_max * 2 * sizeof(double));
    L.off_counts = off; off += mpc_align(64 + (int64_t)(L.nimg > 0 ? L.nimg : 1) * sizeof(float));
    const int64_t bt = (int64_t)(s->B > 0 ? s->B : 1) * s->nb;
    const int km = mpc_knn_margin(s);
    const int64_t hb = s->hq + 2 * km, wb = s->wq + 2 * km, Gb = hb * wb;
    const int64_t ktiles = mpc_knn_tiles(s);
    L.off_cell_start = off; off += mpc_align(bt * (Gb + 1) * sizeof(uint16_t));       // (knn_device.h: knn_cs_t)
    L.off_knn_sat = off;    off += mpc_align(bt * (hb + 1) * (wb + 1) * sizeof(uint16_t));
    L.off_spos = off;       off += mpc_align(bt * (int64_t)s->n * 2 * sizeof(float));
    L.off_sidx = off;       off += mpc_align(bt * (int64_t)s->n * sizeof(uint16_t));               // (knn_idx_t)
    L.off_knn_tmp_g = off;  off += mpc_align(bt * (int64_t)s->n * s->T * 2 * sizeof(float));
    L.off_knn_tmp_a = off;  off += mpc_align(bt * (int64_t)s->n * 2 * sizeof(float));
    L.off_knn_cursor = off; off += mpc_knn_big_sort(s) ? mpc_align(bt * Gb * sizeof(int32_t)) : 0;
    L.off_knn_reach = off;  off += mpc_align(bt * ktiles * sizeof(float));
    L.off_knn_fail = off;   off += mpc_align((1 + bt * (int64_t)L.G) * sizeof(int32_t));
    L.off_knn_retry = off;  off += mpc_align((1 + bt * (int64_t)mpc_cdiv(s->wq, 2) * mpc_cdiv(s->hq, 128)) * sizeof(int32_t));
    const int64_t fitems = (int64_t)mpc_cdiv(s->wq, 2) * mpc_cdiv(s->hq, 128);       // (knn_device.h: KNN_FAR_WS x KNN_FAR_TH blocks of queries)
    L.off_knn_farstrip = off; off += mpc_align((1 + bt * fitems) * sizeof(int32_t));
    L.off_knn_ftlist = off; off += mpc_knn_uses_far_list(s) ? mpc_align((1 + bt * ktiles) * sizeof(int32_t)) : 0;
    L.off_knn_ftbits = off; off += mpc_knn_uses_far_list(s) ? mpc_align(bt * ((ktiles + 31) / 32) * sizeof(int32_t)) : 0;
    L.off_knn_chord = off;  off += mpc_align(1024);
    L.off_knn_again = off;  off += mpc_align(2 * bt * (int64_t)s->hq * ((s->wq + 31) / 32) * sizeof(int32_t));      // (`again` and `grow`)
    L.off_knn_far = off;    off += mpc_knn_uses_far_list(s) ? mpc_align(bt * (1 + (int64_t)L.G) * sizeof(int32_t)) : 0;
    // event partition of the LDS-tiled path (events.hip): strips sized to ~150 KB of 64-bit accumulators
    const int64_t lds_budget = 150 * 1024;
    L.strip_rows = (int)(lds_budget / ((int64_t)s->W * 8));
    if (L.strip_rows > s->H) L.strip_rows = s->H;
    static const int64_t cstrip_kb = getenv("MPC_EV_CSTRIP_KB") ? atoll(getenv("MPC_EV_CSTRIP_KB")) : 48;      // (tuning)
    L.cstrip_rows = (int)((cstrip_kb * 1024) / ((int64_t)s->wq * 16));      // ~3 workgroups of the backward per CU
    if (L.cstrip_rows < 1) L.cstrip_rows = (int)(lds_budget / ((int64_t)s->wq * 16));
    if (L.cstrip_rows > s->hq) L.cstrip_rows = s->hq;
    if (L.strip_rows > 0 && L.cstrip_rows > 0 && s->T == 1 && s->B > 0 && !(s->flags & MPC_F_ATOMIC_PATH)) {
        L.n_strips = mpc_cdiv(s->H, L.strip_rows);
        {
            // k_iwe_accum keeps a strip of 64-bit accumulators in LDS: one 1024-thread workgroup per CU with the largest strip
            // that fits, two with strips of half that height.  Its time is (rounds of workgroups) x (rows of a strip + a fixed
            // part per workgroup); thinner strips also duplicate more records (an event votes into two rows: 1 / rows of them
            // straddle a strip border).  Among 1x .. 3x the smallest strip count the cheapest by that model is taken -- measured
            // at C3 (14 x 2 images): 16 strips of 30 rows (448 workgroups = 1.75 rounds) 23.6 us, 18 of 27 (1.97 rounds) 22.3,
            // 36 of 14 (two per CU, 1.97 rounds) 19.9, 54 of 9: 20.8 with k_ev_bin + 1.7; at B = 1 thin strips are the parallelism
            // there is (C4: 16 strips 15.3 us, 32: 10.6).  MPC_EV_STRIPS=<n> forces a count (tuning).
            // (per device: a process may drive GPUs with different CU counts; -1 = not asked yet)
            static std::atomic<int> ncu_of[64];
            static std::atomic<bool> ncu_init{false};
            if (!ncu_init.exchange(true)) for (auto &v : ncu_of) v.store(0);
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
            int ncu = ncu_of[dev & 63].load();
            if (ncu <= 0) {
                if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) {
                    (void)hipGetLastError();
                    ncu = 256;
                }
                ncu_of[dev & 63].store(ncu);
            }
            static const int forced = getenv("MPC_EV_STRIPS") ? atoi(getenv("MPC_EV_STRIPS")) : 0;
            const int n0 = L.n_strips;
            auto cost_of = [&](int n) {
                const int rows = mpc_cdiv(s->H, n);
                const int64_t lds = (int64_t)rows * s->W * 8 + 1280;
                const int wpc = lds * 2 <= 160 * 1024 ? 2 : 1;
                const int64_t rounds = mpc_cdiv((int64_t)s->B * L.P * n, (int64_t)ncu * wpc);
                return (double)rounds * (rows + 16) * (1.0 + 1.0 / rows);
            };
            int best = n0;
            double best_cost = cost_of(n0);
            // (every forward bucket is sized for the worst case -- all events of its polarity block: more strips, more committed
            // memory.  Beyond 1.5 GB of forward records the smallest count stays: C3 1.2 GB at 35 strips, 0.5 at 16)
            const int64_t mpol_ = (L.P == 2) ? (s->Mp > s->M - s->Mp ? s->Mp : s->M - s->Mp) : s->M;
            for (int n = n0 + 1; n <= 3 * n0 && mpc_cdiv(s->H, n) >= 8; ++n) {
                if ((int64_t)s->B * L.P * n * (mpol_ > 0 ? mpol_ : 1) * 12 > (int64_t)1536 << 20) break;
                const double cost = cost_of(n);
                if (cost < best_cost * 0.999) { best_cost = cost; best = n; }
            }
            if (forced >= n0 && forced <= s->H) best = forced;
            L.n_strips = best;
        }
        L.strip_rows = mpc_cdiv(s->H, L.n_strips);          // equalise the strips
        L.n_cstrips = mpc_cdiv(s->hq, L.cstrip_rows);
        L.cstrip_rows = mpc_cdiv(s->hq, L.n_cstrips);
        L.nfb = s->B * L.P * L.n_strips;
        L.nbb = s->B * s->nb * L.n_cstrips;
        // FORWARD buckets (image strip of the WARPED position: flow dependent, cannot be sized ahead): a bucket holds
        // whatever can reach it -- every event of a polarity block may vote into one image strip.  This memory is
        // committed (the caller's torch.empty is a hipMalloc), nfb * fcap * 12 bytes: 1.2 GB at C3 (35 strips).
        // BACKWARD buckets (the event's own LUT cell: flow independent): none for a forward-only call or bucket-ordered
        // events (MPC_F_NO_BWD_RECORDS: the backward reads the event rows themselves).  Otherwise either every bucket holds
        // all M rows of its sample (nbb * M * 16 bytes: 4.7 GB at C3), or -- where that exceeds MPC_EV_EXACT_ABOVE_MB
        // (default 6144: 2 % of the 288 GB) -- the buckets are sized EXACTLY by a counting pass over the events
        // (ev_count_device.h), M records per sample in all, at the price of that pass (+11 us per C3 step when forced).
        const int64_t mpol = (L.P == 2) ? (s->Mp > s->M - s->Mp ? s->Mp : s->M - s->Mp) : s->M;
        L.fcap = (int)(mpol > 0 ? mpol : 1);
        L.bcap = (int)(s->M > 0 ? s->M : 1);                 // records of one bucket, or (b_exact) of one SAMPLE
        static const int64_t exact_above = (getenv("MPC_EV_EXACT_ABOVE_MB") ? atoll(getenv("MPC_EV_EXACT_ABOVE_MB")) : 6144) << 20;
        L.b_exact = ((int64_t)L.nbb * L.bcap * 16 > exact_above) ? 1 : 0;
        L.off_fcount = off; off += mpc_align((int64_t)(L.nfb + 3 * L.nbb + 8) * sizeof(int32_t));      // fill counters, marker, capacities, first records
        L.off_frec = off;   off += mpc_align((int64_t)L.nfb * L.fcap * 12);
        L.off_brec = off;   off += (s->flags & MPC_F_NO_BWD_RECORDS) ? 0 : mpc_align((int64_t)(L.b_exact ? s->B : L.nbb) * L.bcap * 16);
    } else {
        L.strip_rows = L.cstrip_rows = 0;
    }
    L.total = off;
    return L;
}

extern "C" int64_t mpc_knn_state_floats(const mpc_shape *s) {
    if (!s) { mpc_set_error("mpc_knn_state_floats: null shape"); return MPC_E_NULL; }
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    const int64_t bt = (int64_t)s->B * s->nb;
    return 3 * bt * s->hq * s->wq + bt * mpc_knn_tiles(s) * 5;
}

extern "C" int64_t mpc_knn_fail_list_offset(const mpc_shape *s) {
    if (!s) { mpc_set_error("mpc_knn_fail_list_offset: null shape"); return MPC_E_NULL; }
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    return mpc_layout(s).off_knn_fail;
}

// Diagnostics (tools/fail_probe.py): byte offsets, inside the workspace, of the work lists of the KNN forward --
// out[0] retry, [1] farstrip, [2] again map, [3] grow map, [4] far lists, [5] far tile list; -1 where a list does not exist.
extern "C" int mpc_knn_list_offsets(const mpc_shape *s, int64_t *out) {
    if (!s || !out) { mpc_set_error("mpc_knn_list_offsets: null argument"); return MPC_E_NULL; }
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    const mpc_ws_layout L = mpc_layout(s);
    const int64_t bt = (int64_t)(s->B > 0 ? s->B : 1) * s->nb;
    out[0] = L.off_knn_retry; out[1] = L.off_knn_farstrip; out[2] = L.off_knn_again;
    out[3] = L.off_knn_again + bt * (int64_t)s->hq * ((s->wq + 31) / 32) * 4;
    out[4] = mpc_knn_uses_far_list(s) ? L.off_knn_far : -1; out[5] = mpc_knn_uses_far_list(s) ? L.off_knn_ftlist : -1;
    return 0;
}

extern "C" int64_t mpc_knn_tail_counters_offset(const mpc_shape *s) {
    if (!s) { mpc_set_error("mpc_knn_tail_counters_offset: null shape"); return MPC_E_NULL; }
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    return mpc_layout(s).off_knn_chord + 512;          // (knn_device.h: knn_marked_count, knn_late_count, knn_tail_done)
}

extern "C" int64_t mpc_workspace_bytes(const mpc_shape *s) {
    if (!s) { mpc_set_error("mpc_workspace_bytes: null shape"); return MPC_E_NULL; }
    int rc = mpc_validate_shape(s);
    if (rc) return rc;
    return mpc_layout(s).total;
}

// ---- A4: FocusLoss.calc and its backward as one call each (reference src/losses/focus.py:66-113) -----------------
extern "C" int mpc_focus_fwd(const mpc_shape *s, const mpc_focus_buffers *io, void *ws, void *stream) {
    MPC_CHECK_ARG(s && io && ws, MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(io->traj && io->flow_lut && io->knn_state && io->iwe_raw && io->iwe_blur && io->scal, MPC_E_NULL, "null buffer");
    // the KNN forward's kernels also zero the event bucket counters and (unless the backward reads the rows themselves) count
    // the rows per backward bucket: `done` says what of that happened (bit 0 zeroed, bit 1 counted)
    const bool rec_bwd = io->grad_iwe && !io->event_offsets && !(s->flags & MPC_F_NO_BWD_RECORDS);
    int done = 0;
    int rc = mpc_knn_lut_fwd_ex(s, io->traj, io->flow_lut, io->flow_next, io->knn_state, nullptr, ws, stream, 1, rec_bwd ? io->events : nullptr, &done);
    if (rc) return rc;
    int s_nimg = 0, s_C = 0;
    if (io->smooth_weight > 0.f) {
        const bool on_next = (s->flags & MPC_F_WANT_NEXT) != 0;
        const float *field = on_next ? io->flow_next : io->flow_lut;
        s_nimg = on_next ? s->B * (s->nb - 1) : s->B * s->nb;
        s_C = on_next ? 2 : 2 * s->T;
        if (s_nimg > 0) {
            if ((rc = mpc_lut_smooth(s, field, s_nimg, s_C, io->smooth_weight, io->smooth_grad, ws, stream))) return rc;
        } else s_C = 0;
    }
    // (the bucket counters were zeroed by the first kernel of the KNN forward, unless the event path is not the tiled one)
    {
        // ordered events: no record per event for the backward (it reads the rows themselves)
        mpc_shape sf = *s;
        if (io->event_offsets) sf.flags |= MPC_F_NO_BWD_RECORDS;
        if ((rc = mpc_event_splat_fwd_ex(&sf, io->events, io->flow_lut, io->t_ref, io->iwe_raw, ws, stream, done, io->event_offsets))) return rc;
    }
    if ((rc = mpc_contrast_fwd(s, io->iwe_raw, io->iwe_blur, io->grad_iwe, ws, stream))) return rc;
    return mpc_finalize_ex(s, s_nimg, s_C, io->smooth_weight, io->scal, io->scal_out, ws, stream);
}

extern "C" int mpc_focus_bwd(const mpc_shape *s, const mpc_focus_buffers *io, const float *grad_out,
                             float *grad_lut_scratch, float *grad_next_scratch, float *grad_traj, void *ws, void *stream) {
    MPC_CHECK_ARG(s && io && ws && grad_lut_scratch && grad_traj, MPC_E_NULL, "null argument");
    MPC_CHECK_ARG(io->grad_iwe, MPC_E_NULL, "mpc_focus_fwd was called without grad_iwe (forward only)");
    const bool on_next = (s->flags & MPC_F_WANT_NEXT) != 0;
    const bool smooth = io->smooth_weight > 0.f && io->smooth_grad != nullptr;
    // the smoothness gradient on flow_to_tref is folded into the event backward; on flow_to_next it is a separate
    // gradient of the KNN backward, scaled by grad_out
    int reach_done = 0;       // (the event backward's kernel computes the tile reaches of the KNN backward on the side)
    int rc = mpc_event_splat_bwd_job(s, io->events, io->event_offsets, io->flow_lut, io->t_ref, io->grad_iwe, io->scal, grad_out,
                                     grad_lut_scratch, (smooth && !on_next) ? io->smooth_grad : nullptr, ws, stream, io->knn_state, &reach_done);
    if (rc) return rc;
    // (smoothness on flow_to_next: dL/dflow_next = grad_out * the saved smoothness gradient -- the KNN backward's kernels multiply as
    // they read it, no pass over the field)
    const bool next_grad = smooth && on_next && s->nb > 1;
    MPC_CHECK_ARG(!next_grad || !grad_out || grad_next_scratch, MPC_E_NULL, "grad_next_scratch is null");
    return mpc_knn_lut_bwd_ex(s, io->traj, grad_lut_scratch, next_grad ? io->smooth_grad : nullptr, io->knn_state, grad_traj, ws, stream, reach_done,
                              next_grad ? grad_out : nullptr, grad_next_scratch);
}
