// C ABI of libscfgp_hip.so (declared in include/scfgp_hip.h): context, device memory,
// the staged evaluation  pass1 | factor | pass2 | adjoint | pass3 | finish  and predict.
#include "../../include/scfgp_hip.h"
#include "kernels.h"

#include <dlfcn.h>
// RCCL: types and prototypes only -- the library is looked up at run time (scfgp_comm_init), no link-time dependency.  A ROCm
// install without the RCCL development headers still builds: the handful of declarations used here are then spelled out
// (they are RCCL's public, NCCL-compatible ABI).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
ncclResult_t ncclCommDestroy(ncclComm_t);
const char* ncclGetErrorString(ncclResult_t);
}
#endif

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define HIPCHK(ctx, call)                                                                           \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
            return SCFGP_EHIP;                                                                      \
        }                                                                                           \
    } while (0)

static constexpr int XT = 128;              // tile of the X~^T Zbar product and padding unit of J (kernels_sweep.hip)
static constexpr int64_t PRED_ROWS = 32768; // predict processes test rows in chunks of this size
// Thresholds of the automatic precision escalation and the error model behind them (profiles/r03_c3_owner.md): with the
// fp32 Gram products (4096-row flush interval, relative error ~6e-8) the relative error of alpha / Li, and of the frequency
// gradient blocks, against fp64 mode is about SCFGP_ERR_PER_COND times the condition estimate (measured 7e-8 .. 2e-6 times
// the estimate at estimates 4 .. 6.4e3: the estimate is a lower bound of cond_2 of varying tightness).  Level 1 keeps the predicted alpha error under 3e-6 (north star: 1e-5), level 2
// the predicted gradient error under 3e-4 (SURVEY App. E acceptance: 1e-3 per block).
#ifndef SCFGP_COND_THRESHOLD
#define SCFGP_COND_THRESHOLD 10.0
#endif
#ifndef SCFGP_COND_THRESHOLD_W
#define SCFGP_COND_THRESHOLD_W 1000.0
#endif
#ifndef SCFGP_ERR_PER_COND
#define SCFGP_ERR_PER_COND 3.0e-7
#endif
static void derive_geom(Geom& g, int D, int S, int M);

thread_local bool g_scfgp_capturing = false;

// roctx ranges around every stage (visible in `rocprofv3 --marker-trace`): OPT-IN (environment SCFGP_ROCTX=1 at context
// creation or option "roctx"); only then is the profiler's roctx library looked up, at run time, so the product has no
// link-time dependency on the profiler SDK and an ordinary process never loads it
struct Roctx {
    int (*push)(const char*) = nullptr; int (*pop)() = nullptr;
    Roctx() {
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_LOCAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_LOCAL);
        if (!h) return;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
static Roctx& roctx() { static Roctx r; return r; }

// RCCL, resolved at run time the first time a communicator is asked for: a process that already carries a copy (the host
// framework's, e.g. torch's librccl.so) is served by that copy, otherwise ROCm's is loaded.  Single-GPU users never load it.
struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) get_id = nullptr; decltype(&ncclCommInitRank) init_rank = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr; decltype(&ncclCommDestroy) destroy = nullptr;
    decltype(&ncclGetErrorString) err_string = nullptr;
    bool resolved = false;
    bool bind(void* from) {
        get_id = (decltype(get_id))dlsym(from, "ncclGetUniqueId"); init_rank = (decltype(init_rank))dlsym(from, "ncclCommInitRank");
        all_reduce = (decltype(all_reduce))dlsym(from, "ncclAllReduce"); destroy = (decltype(destroy))dlsym(from, "ncclCommDestroy");
        err_string = (decltype(err_string))dlsym(from, "ncclGetErrorString");
        return get_id && init_rank && all_reduce && destroy && err_string;
    }
    Rccl() {
        // a copy the process already carries, WHEREVER it was loaded from (a host framework bundles its own librccl under its
        // own path and may have loaded it RTLD_GLOBAL): the global symbol scope first, then an already-mapped copy by soname
        // (RTLD_NOLOAD; never promoted to RTLD_GLOBAL), and only then ROCm's copy, privately
        if (dlsym(RTLD_DEFAULT, "ncclAllReduce") && bind(RTLD_DEFAULT)) { resolved = true; return; }
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
        for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        if (bind(h)) resolved = true;
        else { dlclose(h); h = nullptr; }
    }
    bool ok() const { return resolved; }
};
static Rccl& rccl() { static Rccl r; return r; }

struct ProfRec { std::string name; hipEvent_t e0, e1; };

struct scfgp_ctx {
    Geom g{};
    int dtype = 0, device = 0;
    hipStream_t st = nullptr; bool own_stream = false;
    hipStream_t copy_st = nullptr; hipEvent_t ev_factor = nullptr;          // alpha/Li D2H beside pass 2/3 ...
    // rank-S form of the backward projection: exchange 3 = [T~^T Zbar (Spp x Jp) ... | 8 scalars at Dpp*Jp | X~^T U (Dpp x Sq)]
    int lowrank_bwd = -1; int Spp = 0, Sq = 0; bool last_lrb = false;
    bool want_lrb() const {
        // T~ = [X l_F | 1] exists only when the forward projection goes through the S columns (g.lowrank), and U needs room in
        // Phibar's dead sine half
        const bool fits = g.lowrank && g.Jp + (int)round_up(g.S, 64) <= g.Kp;
        return fits && (lowrank_bwd == 1 || (lowrank_bwd < 0 && g.Dp >= 4 * g.Sp));
    }
    double* x3_scalars() { return d_x3 + (int64_t)Dpp * g.Jp; }
    double* xs1() { return d_xp1 + n_pk + g.Kp; }                           // the 8 scalars that close exchange buffers 1 and 2
    double* xs2() { return d_xp2 + n_pk + g.Kp; }
    // Ranks decide together (common.h: XS_*).  `unsettled`: a precision level was raised (or refused) since the last sum over
    // ranks, so the next exchange 1 carries every rank's attainable level and scfgp_factor commits to the lowest before pass 2
    // runs; cap(): the highest level this rank can run at.  test_*: fault injection for the tests (scfgp_set_option).
    bool unsettled = false; int test_deny_level = 0, test_fail_stage = 0;
    int cap() const { return esc_denied > 0 ? esc_denied - 1 : 2; }
    double* x3_xu() { return d_x3 + (int64_t)Dpp * g.Jp + 8; }
    hipEvent_t ev_fence = nullptr;                                          // scfgp_stream_fence
    ncclComm_t comm = nullptr; int comm_ranks = 0, comm_rank = 0;           // scfgp_comm_init: the three sums run inside the library
    double* h_pin = nullptr;                                                // ... through pinned staging (K*K + K doubles)
    int64_t Ncap = 0, Nglobal = 0, Nglobal_full = 0;        // Nglobal_full: n_global given to scfgp_set_data
    bool have_params = false, have_data = false;
    int stage = 0, last_want_grad = 0;
    std::vector<double> h_params;
    // parameters
    double *d_params = nullptr, *d_F = nullptr, *d_Fall = nullptr, *d_Lall = nullptr, *d_Rall = nullptr; Scal* d_sc = nullptr;
    double *d_Tt = nullptr, *p_Tt = nullptr;                     // T~ = X~ Lall of the rank-S projection (rows / predict chunk)
    // dataset store (raw rows as uploaded) and the working set the sweeps run on
    double *d_Xraw = nullptr, *d_yraw = nullptr; int64_t Nstore = 0, store_cap = 0; bool work_full = false;
    int64_t* d_idx = nullptr; int64_t idx_cap = 0;
    // rows
    double *d_Xt = nullptr, *d_y = nullptr, *d_p = nullptr, *d_q = nullptr, *d_mu = nullptr, *d_vpart = nullptr;
    void *d_Phi = nullptr, *d_V = nullptr;
    // compute mode SCFGP_F16X3 (apply_f16.hip): fp32 mode whose two square apply products run as a three-term fp16 split wherever
    // fp32 mode would use its 256-wide LDS-DMA tiles; Phi16 = Phi in plane form (kernels.h), B16 = the K x K operand in plane form
    bool split16 = false; unsigned* d_Phi16 = nullptr; char* d_B16 = nullptr; float* d_f16scale = nullptr;
    // ... and the two Gram products too (gram_f16.hip): pass 1's on Phi16, pass 2's on the plane forms of V and of diag(q) V; block
    // partials of the side vectors; 8 floats of bounds and scales
    unsigned* d_V16g = nullptr; unsigned* d_qV16g = nullptr; double* d_f16side = nullptr; float* d_f16tmp = nullptr;
    int f16gram = 1;                                             // tuning knob "f16_gram": 0 keeps the fp32 Gram in this mode
    bool f16_on() const { return split16 && dma() == 2; }
    // rows per job and per slab set of the fp16 Gram (its fp32 chains end every 512 rows whatever this is): twice the fp32 Gram's flush
    // interval, less where that would leave fewer than 512 jobs = two rounds of the chip
    int64_t f16_chunk() const {
        const int64_t top = gram_chunk > 0 ? 2 * gram_chunk : 8192, fill = g.Np * F16x3Kernels::gram_tiles(g) / 512 / 512 * 512;
        return std::max<int64_t>(std::min(top, fill), 1024);
    }
    bool f16_gram() const { return f16_on() && f16gram && !last_cform; }      // factor form (level 2) keeps its fp32 products
    F16Operands f16ops() const { return F16Operands{d_Phi16, d_B16, d_f16scale, nullptr, d_f16tmp + 5}; }
    // exchange buffers and K-stage
    // exchange buffers xp1/xp2 = [packed lower tiles | vector Kp | 8 scalars], x3 = [X~^T Zbar | 8 scalars];
    // x1/x2 = the same matrices unpacked to full Kp x Kp (+ vector) for the K x K stage
    double *d_xp1 = nullptr, *d_xp2 = nullptr; int64_t n_xp = 0, n_pk = 0;
    double *d_x1 = nullptr, *d_x2 = nullptr, *d_x3 = nullptr; int64_t n_x1 = 0, n_x2 = 0, n_x3 = 0; int Dpp = 0;
    double *d_Li = nullptr, *d_B = nullptr, *d_T1 = nullptr, *d_T2 = nullptr, *d_Abar = nullptr;
    void *d_BT = nullptr, *d_AbarT = nullptr;                      // sweep operands: typed copies of B (or Li) and Abar (or Li^T)
    int apply_dma = -1;                                            // option: -1 = by problem size, 0 off, 1 = 128-wide tiles, 2 = 256-wide (fp32)
    double *d_vecs = nullptr;            // beta, alpha, u, ut, alpha_pred (Kp each)
    double *d_scalars = nullptr, *d_yy = nullptr; int* d_flag = nullptr;
    float* d_ws2 = nullptr;                                        // fp32 mode: packed (weight, side multiplier) rows of the Gram launches (kernels.h)
    double *d_slabs = nullptr; size_t slabs_bytes = 0;
    double *d_partial = nullptr; int64_t n_partial = 0;
    double *d_work = nullptr, *d_grad = nullptr;
    int xs_mode = 0; double* d_xscale = nullptr;                 // X scaler for scfgp_predict_raw (5*D doubles)
    int ys_mode = 0; double* d_yscale = nullptr;                 // y scaler for scfgp_predict_y (5 doubles)
    // predict chunk buffers
    double *p_Xt = nullptr, *p_vpart = nullptr, *p_mupart = nullptr; void *p_Phi = nullptr;
    // on-device optimiser + captured training iteration
    int opt_algo = -1; OptHyper opt_h{}; double *d_opt = nullptr, *d_tctr = nullptr, *d_hist = nullptr; int hist_cap = 0;
    hipGraph_t graph = nullptr; hipGraphExec_t gexec = nullptr; int64_t graph_N = -1; bool in_train = false, warm = false;
    int use_graph = 1;
    // options
    int gram_nsplit = 0, gram_taper = 1, xtz_nsplit = 0; int64_t gram_chunk = 4096;
    RowSplits splits{};
    // Precision escalation (fp32 mode; profiles/r03_c3_owner.md).  The fp32 Gram products carry a relative error
    // of ~6e-8 that reaches alpha / Li (pass 1) and the gradient (pass 2's V^T diag(q) V) multiplied by the condition of A,
    // so the K x K stage's free condition estimate decides how the two TN products are formed:
    //   level 0  both in fp32 MFMA (fp64 across 4096-row chunks)
    //   level 1  pass 1: G = Phi^T Phi and Phi^T y by the fp64 kernels from fp64 features (alpha, Li, cost, mu*, sigma* then
    //            equal fp64 mode's to ~1e-8; alpha and Li bit for bit)
    //   level 2  also pass 2 in its factor form (C = Phi Li^T, v = rowsum(C^2), B W B = Li^T (C^T diag(q) C) Li, V = C Li):
    //            rounding errors amplified by sqrt(cond) instead of cond (gradient blocks)
    // option gram64: 0 never, 1 always level 1, 2 auto (default), 3 always level 2.  Auto is sticky: when the estimate of a
    // finished evaluation asks for a higher level that evaluation is repeated (scfgp_finish returns SCFGP_REDO; scfgp_eval
    // does so itself) and the following ones start at that level; the level drops when the estimate falls below a quarter
    // of the threshold.  Inside one scfgp_train call the level is fixed (the iteration may be a captured graph).
    int gram64 = 2, esc_level = 0, esc_denied = 0; double esc_thr = SCFGP_COND_THRESHOLD, escw_thr = SCFGP_COND_THRESHOLD_W; bool last_used64 = false;
    int last_level = 0; bool cond_valid = false;               // cond_valid: cond[] describes the current parameters and rows
    double cond[3] = {0, 0, 0};                                         // min L_ii^2, max L_ii^2, max_j (A^-1)_jj of the last factorisation
    void* d_Phi64 = nullptr; int64_t phi64_cap = 0; RowSplits splits64{};
    // factor form of pass 2 (C = Phi Li^T, V = C Li: SweepKernels::apply_c): -1 auto (precision level 2), 0 never, 1 always
    int factor_form = -1; bool last_cform = false; void* d_C = nullptr; int64_t c_cap = 0;
    bool want_cform() const { return factor_form == 1 || (factor_form < 0 && level() >= 2); }
    int level() const { return dtype != SCFGP_F32 || gram64 == 0 ? 0 : (gram64 == 1 ? 1 : (gram64 == 3 ? 2 : esc_level)); }
    bool use64() const { return level() >= 1; }
    double cond_est() const { return cond[1] * cond[2]; }
    // predicted relative error of alpha / Li behind an fp32 Gram: SCFGP_ERR_PER_COND x estimate.  The estimate is a lower bound
    // of cond_2(A) whose tightness varies (cond_2 / estimate: 30 .. 330 over the measured cases), so the prediction is an order
    // of magnitude: measured error / prediction lies in 0.2 .. 7 (profiles/r03_c3_owner.md)
    double alpha_err_fp32() const { return SCFGP_ERR_PER_COND * cond_est(); }
    int want_level(double est, double slack) const { return est > escw_thr * slack ? 2 : (est > esc_thr * slack ? 1 : 0); }
    // profiling
    bool roctx_on = false;
    bool prof = false; std::vector<ProfRec> recs; std::vector<hipEvent_t> pool; size_t pool_used = 0;
    std::string err;

    double* beta() { return d_vecs; }
    double* alpha() { return d_vecs + g.Kp; }
    // the apply products by LDS-DMA (apply.hip: apply_dma_kernel).  Auto: whenever the column plan has 128-wide tiles (K > 256)
    // and there are >= 16384 rows -- measured at K = 544 and 992, 16384 .. 100000 rows, both dtypes: 128-wide LDS-DMA tiles are
    // 4-13 % ahead of the loader-staged ones per product, 256-wide ones behind at these sizes (profiles/r04_tuning.md); fp32
    // 256-wide tiles from K >= 1024 and 65536 rows (equal to 128-wide at H since both are pipelined, 2/3 of the fabric traffic).
    // Below that the loader-staged tiles, whose single 64-wide launch per product matters more there
    int dma() const {
        const int auto_dma = g.K > 256 && g.Np >= 16384 ? (dtype == SCFGP_F32 && g.K >= 1024 && g.Np >= 65536 ? 2 : 1) : 0;
        return apply_dma < 0 ? auto_dma : apply_dma;
    }
    double* u() { return d_vecs + 2 * g.Kp; }
    double* ut() { return d_vecs + 3 * g.Kp; }
    double* alpha_pred() { return d_vecs + 4 * g.Kp; }
    size_t tsize() const { return dtype == SCFGP_F32 ? 4 : 8; }
    double h_scale() const { return std::exp(h_params[1]) * std::sqrt(2.0 / g.M); }      // s = e^b sqrt(2/M) from the host copy
    KStage kstage() {
        KStage k; k.K = g.K; k.Kp = g.Kp; k.A = d_x1; k.Li = d_Li; k.B = d_B; k.T1 = d_T1; k.T2 = d_T2;
        k.g = d_x1 + (int64_t)g.Kp * g.Kp; k.beta = beta(); k.alpha = alpha(); k.h = d_x2 + (int64_t)g.Kp * g.Kp;
        k.u = u(); k.ut = ut(); k.scalars = d_scalars; k.flag = d_flag;
        return k;
    }
};

struct ProfScope {
    scfgp_ctx* c; size_t idx = (size_t)-1; bool ranged = false;
    ProfScope(scfgp_ctx* c_, const char* name) : c(c_) {
        if (g_scfgp_capturing) return;
        if (c->roctx_on && roctx().push) { roctx().push(name); ranged = true; }
        if (!c->prof) return;
        auto get = [&]() {
            if (c->pool_used == c->pool.size()) { hipEvent_t e; hipEventCreate(&e); c->pool.push_back(e); }
            return c->pool[c->pool_used++];
        };
        ProfRec r; r.name = name; r.e0 = get(); r.e1 = get();
        hipEventRecord(r.e0, c->st);
        c->recs.push_back(r); idx = c->recs.size() - 1;
    }
    ~ProfScope() {
        if (idx != (size_t)-1) hipEventRecord(c->recs[idx].e1, c->st);
        if (ranged) roctx().pop();
    }
};

template <typename P> static int dmalloc(scfgp_ctx* c, P** p, size_t bytes) {
    HIPCHK(c, hipMalloc((void**)p, bytes ? bytes : 16));
    return SCFGP_OK;
}
template <typename P> static void dfree(P*& p) { if (p) { hipFree((void*)p); p = nullptr; } }
// call-scoped device buffer: released on every return path
struct DevTmp {
    double* p = nullptr;
    ~DevTmp() { if (p) hipFree(p); }
    operator double*() const { return p; }
};

// Row splits of the X~^T Zbar grid (few output tiles, equally long workgroups, two per CU): ONE round of the 512 resident
// workgroups.  The former "about 1152 workgroups" were 2.25 rounds at the headline shape (9 tiles x 128 splits) -- a third
// round a quarter full: 4.00 ms against 3.24 at 56 splits = 504 workgroups, 3.28-3.32 at two to six full rounds
// (profiles/r05_tuning.md).
static int xtz_split(int ntiles, int64_t Np) {
    int64_t s = std::max(1, 512 / ntiles);
    int64_t smax = std::max<int64_t>(Np / 2048, 1);
    if (ntiles * smax < 256) smax = std::max<int64_t>(Np / 64, 1);      // small problems: short dependent k-loops instead of few long ones
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    return (int)s;
}

static void free_rows(scfgp_ctx* c) {
    dfree(c->d_Xt); dfree(c->d_y); dfree(c->d_p); dfree(c->d_q); dfree(c->d_ws2); dfree(c->d_mu); dfree(c->d_vpart); dfree(c->d_Tt);
    dfree(c->d_Phi); dfree(c->d_V); dfree(c->d_Phi16); dfree(c->d_V16g); dfree(c->d_qV16g); dfree(c->d_f16side); dfree(c->d_slabs);
    c->Ncap = 0; c->slabs_bytes = 0;
}

// optional row buffers (fp64 features of an escalated pass 1, C of the factor form): sized
// for the current working set, allocated HERE and never inside a pass -- a pass may be running under graph capture
static int ensure_aux_rows(scfgp_ctx* c) {
    const Geom& g = c->g;
    if (c->use64() && c->phi64_cap < g.Np) {
        dfree(c->d_Phi64); c->phi64_cap = 0;
        if (int rc = dmalloc(c, &c->d_Phi64, sizeof(double) * g.Np * g.Kp)) return rc;
        HIPCHK(c, hipMemsetAsync(c->d_Phi64, 0, sizeof(double) * g.Np * g.Kp, c->st));      // columns >= K stay zero
        c->phi64_cap = g.Np;
    }
    if (c->want_cform() && c->c_cap < g.Np) {
        dfree(c->d_C); c->c_cap = 0;
        if (int rc = dmalloc(c, &c->d_C, c->tsize() * g.Np * g.Kp)) return rc;
        HIPCHK(c, hipMemsetAsync(c->d_C, 0, c->tsize() * g.Np * g.Kp, c->st));
        c->c_cap = g.Np;
    }
    return SCFGP_OK;
}

// (re)allocate every buffer whose size depends on the number of local rows
static int ensure_rows(scfgp_ctx* c, int64_t N) {
    Geom& g = c->g;
    if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }      // captured pointers/sizes may change
    if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
    const int64_t Np = round_up(N > 0 ? N : 1, 256);
    g.N = N; g.Np = Np;
    const int nts = g.Kp / g.tile, ntiles = nts * (nts + 1) / 2;
    const int ntx = ((g.Dp + XT - 1) / XT) * (g.Jp / XT);
    const int gjobs = c->dtype == SCFGP_F32 ? SweepKernels<float>::gram_jobs(g) : SweepKernels<double>::gram_jobs(g);
    c->splits = gram_row_splits(gjobs, Np, c->dtype == SCFGP_F32, c->gram_nsplit, c->gram_taper);
    int gs = c->splits.nsplit;
    if (c->dtype == SCFGP_F32 && c->gram64 != 0) {              // the fp64 job list of an escalated pass 1 has its own row splits
        c->splits64 = gram_row_splits(SweepKernels<double>::gram_jobs(g), Np, false, c->gram_nsplit, c->gram_taper);
        gs = std::max(gs, c->splits64.nsplit);
    }
    const int xs = c->xtz_nsplit > 0 ? (int)std::min<int64_t>(c->xtz_nsplit, Np / 64) : xtz_split(ntx, Np);
    // Gram slabs are followed by the per-split side-vector partials (gs x Kp)
    // the two row-contracted products of the rank-S backward projection (pass3): (Spp x Jp) and (Dpp x Sq) tile grids
    const int nt1 = (c->Spp / XT) * (g.Jp / XT), nt2 = (c->Dpp / XT) * (c->Sq / XT);
    const size_t lrb = std::max<size_t>((size_t)xtz_split(nt1, Np) * nt1, (size_t)xtz_split(nt2, Np) * nt2);
    if (c->split16) gs = std::max(gs, F16x3Kernels::gram_chunks(g, c->f16_chunk()));       // the f16x3 Gram: one set of slabs per row chunk
    const size_t need = sizeof(double) * std::max<size_t>((size_t)gs * ntiles * g.tile * g.tile + (size_t)gs * g.Kp,
                                                          std::max<size_t>((size_t)xs * ntx, lrb) * XT * XT);
    if (need > c->slabs_bytes) {
        dfree(c->d_slabs);
        if (int rc = dmalloc(c, &c->d_slabs, need)) return rc;
        c->slabs_bytes = need;
    }
    if (int rc = ensure_aux_rows(c)) return rc;
    if (Np <= c->Ncap) return SCFGP_OK;
    dfree(c->d_Xt); dfree(c->d_y); dfree(c->d_p); dfree(c->d_q); dfree(c->d_ws2); dfree(c->d_mu); dfree(c->d_vpart); dfree(c->d_Tt);
    dfree(c->d_Phi); dfree(c->d_V); dfree(c->d_Phi16); dfree(c->d_V16g); dfree(c->d_qV16g); dfree(c->d_f16side);
    c->Ncap = 0;
    const size_t ts = c->tsize();
    int rc;
    if ((rc = dmalloc(c, &c->d_Xt, sizeof(double) * Np * g.Dp))) return rc;
    if (g.lowrank && (rc = dmalloc(c, &c->d_Tt, sizeof(double) * Np * g.Sp))) return rc;
    if ((rc = dmalloc(c, &c->d_y, sizeof(double) * Np))) return rc;
    if ((rc = dmalloc(c, &c->d_p, sizeof(double) * Np))) return rc;
    if ((rc = dmalloc(c, &c->d_q, sizeof(double) * Np))) return rc;
    if (c->dtype == SCFGP_F32 && (rc = dmalloc(c, &c->d_ws2, sizeof(float) * 2 * (Np + 32)))) return rc;
    if ((rc = dmalloc(c, &c->d_mu, sizeof(double) * Np * (g.Kp / 64)))) return rc;           // mupart, like vpart
    if ((rc = dmalloc(c, &c->d_vpart, sizeof(double) * Np * (g.Kp / 64)))) return rc;          // <= one entry per 64 columns
    if ((rc = dmalloc(c, &c->d_Phi, ts * Np * g.Kp))) return rc;
    if ((rc = dmalloc(c, &c->d_V, ts * Np * g.Kp))) return rc;
    if (c->split16) {
        const size_t plane = sizeof(unsigned) * Np * g.Kp + F16x3Kernels::F16_PAD;
        if ((rc = dmalloc(c, &c->d_Phi16, plane)) || (rc = dmalloc(c, &c->d_V16g, plane)) || (rc = dmalloc(c, &c->d_qV16g, plane))) return rc;
        if ((rc = dmalloc(c, &c->d_f16side, sizeof(double) * F16x3Kernels::side_blocks(g) * g.Kp))) return rc;
        HIPCHK(c, hipMemsetAsync(c->d_Phi16, 0, plane, c->st));           // the padding is read (into tiles nobody stores): keep it finite
        HIPCHK(c, hipMemsetAsync(c->d_V16g, 0, plane, c->st));
        HIPCHK(c, hipMemsetAsync(c->d_qV16g, 0, plane, c->st));
    }
    HIPCHK(c, hipMemsetAsync(c->d_Phi, 0, ts * Np * g.Kp, c->st));       // padding columns >= K stay zero forever
    HIPCHK(c, hipMemsetAsync(c->d_V, 0, ts * Np * g.Kp, c->st));         // columns >= K are never written
    c->Ncap = Np;
    return SCFGP_OK;
}

extern "C" int scfgp_create(scfgp_ctx** out, int D, int S, int M, int dtype, int device, void* stream) {
    if (!out || D < 1 || S < 1 || M < 1 || (dtype != SCFGP_F64 && dtype != SCFGP_F32 && dtype != SCFGP_F16X3)) return SCFGP_EARG;
    scfgp_ctx* c = new scfgp_ctx();
    *out = c;
    c->split16 = dtype == SCFGP_F16X3;                            // fp32 mode in everything but the two square apply products
    if (c->split16) dtype = SCFGP_F32;
    c->dtype = dtype; c->device = device;
    Geom& g = c->g;
    derive_geom(g, D, S, M);
    if (const char* e = getenv("SCFGP_LOWRANK")) g.lowrank = atoi(e) != 0;                       // tuning override
    if (const char* e = getenv("SCFGP_ROCTX")) c->roctx_on = atoi(e) != 0;
    c->Dpp = (int)round_up(g.Dp, XT);
    HIPCHK(c, hipSetDevice(device));
    if (stream) c->st = (hipStream_t)stream;
    else { HIPCHK(c, hipStreamCreate(&c->st)); c->own_stream = true; }
    HIPCHK(c, hipStreamCreateWithFlags(&c->copy_st, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_factor, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_fence, hipEventDisableTiming));
    const int64_t Kp = g.Kp, K2 = Kp * Kp;
    c->Spp = (int)round_up(g.Sp, XT); c->Sq = (int)round_up(g.S, XT);
    c->n_x1 = K2 + Kp + 8; c->n_x2 = K2 + Kp + 8; c->n_x3 = (int64_t)std::max(c->Dpp, c->Spp) * g.Jp + 8 + (int64_t)c->Dpp * c->Sq;
    { const int64_t nts = Kp / g.tile; c->n_pk = nts * (nts + 1) / 2 * g.tile * g.tile; c->n_xp = c->n_pk + Kp + 8; }
    int rc;
    if ((rc = dmalloc(c, &c->d_params, sizeof(double) * g.P))) return rc;
    if ((rc = dmalloc(c, &c->d_F, sizeof(double) * D * M))) return rc;
    if ((rc = dmalloc(c, &c->d_Fall, sizeof(double) * g.Dp * g.Jp))) return rc;
    if ((rc = dmalloc(c, &c->d_Lall, sizeof(double) * g.Dp * round_up(g.Sp, 64)))) return rc;
    if ((rc = dmalloc(c, &c->d_Rall, sizeof(double) * g.Sp * g.Jp))) return rc;
    if ((rc = dmalloc(c, &c->d_sc, sizeof(Scal)))) return rc;
    if ((rc = dmalloc(c, &c->d_xp1, sizeof(double) * c->n_xp))) return rc;
    if ((rc = dmalloc(c, &c->d_xp2, sizeof(double) * c->n_xp))) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_xp1, 0, sizeof(double) * c->n_xp, c->st));
    HIPCHK(c, hipMemsetAsync(c->d_xp2, 0, sizeof(double) * c->n_xp, c->st));
    if ((rc = dmalloc(c, &c->d_x1, sizeof(double) * c->n_x1))) return rc;
    if ((rc = dmalloc(c, &c->d_x2, sizeof(double) * c->n_x2))) return rc;
    if ((rc = dmalloc(c, &c->d_x3, sizeof(double) * c->n_x3))) return rc;
    if ((rc = dmalloc(c, &c->d_Li, sizeof(double) * K2))) return rc;
    if ((rc = dmalloc(c, &c->d_B, sizeof(double) * K2))) return rc;
    if ((rc = dmalloc(c, &c->d_T1, sizeof(double) * K2))) return rc;
    if ((rc = dmalloc(c, &c->d_T2, sizeof(double) * K2))) return rc;
    if ((rc = dmalloc(c, &c->d_Abar, sizeof(double) * K2))) return rc;
    if ((rc = dmalloc(c, &c->d_BT, c->tsize() * K2))) return rc;        // sweep operands: typed, padding zeroed
    if ((rc = dmalloc(c, &c->d_AbarT, c->tsize() * K2))) return rc;
    if (c->split16) {
        if ((rc = dmalloc(c, &c->d_B16, 4 * (size_t)K2))) return rc;
        if ((rc = dmalloc(c, &c->d_f16scale, sizeof(float) * 4))) return rc;
        if ((rc = dmalloc(c, &c->d_f16tmp, sizeof(float) * 8))) return rc;
    }
    if ((rc = dmalloc(c, &c->d_vecs, sizeof(double) * 5 * Kp))) return rc;
    if ((rc = dmalloc(c, &c->d_scalars, sizeof(double) * 32))) return rc;
    if ((rc = dmalloc(c, &c->d_yy, sizeof(double) * 8))) return rc;
    if ((rc = dmalloc(c, &c->d_flag, sizeof(int) * 16))) return rc;         // [0..3] not-positive-definite flag, [8..15] the Gram's queue heads
    c->n_partial = 16384;                                             // block partials of the scalar reductions
    if ((rc = dmalloc(c, &c->d_partial, sizeof(double) * c->n_partial))) return rc;
    if ((rc = dmalloc(c, &c->d_work, sizeof(double) * (4 * D + 4 + (int64_t)D * M)))) return rc;
    if ((rc = dmalloc(c, &c->d_grad, sizeof(double) * g.P))) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_x1, 0, sizeof(double) * c->n_x1, c->st));
    HIPCHK(c, hipMemsetAsync(c->d_x2, 0, sizeof(double) * c->n_x2, c->st));
    HIPCHK(c, hipMemsetAsync(c->d_x3, 0, sizeof(double) * c->n_x3, c->st));
    HIPCHK(c, hipMemsetAsync(c->d_Li, 0, sizeof(double) * K2, c->st));          // its upper blocks stay zero for good (kstage_factor)
    HIPCHK(c, hipMemsetAsync(c->d_vecs, 0, sizeof(double) * 5 * Kp, c->st));
    HIPCHK(c, hipMemsetAsync(c->d_scalars, 0, sizeof(double) * 32, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return SCFGP_OK;
}

extern "C" void scfgp_destroy(scfgp_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->st) hipStreamSynchronize(c->st);
    free_rows(c);
    dfree(c->d_Phi64); dfree(c->d_C); dfree(c->d_Xraw); dfree(c->d_yraw); dfree(c->d_idx); dfree(c->d_xscale); dfree(c->d_yscale);
    dfree(c->d_params); dfree(c->d_F); dfree(c->d_Fall); dfree(c->d_Lall); dfree(c->d_Rall); dfree(c->d_sc); dfree(c->p_Tt);
    dfree(c->d_xp1); dfree(c->d_xp2); dfree(c->d_x1); dfree(c->d_x2); dfree(c->d_x3); dfree(c->d_Li); dfree(c->d_B); dfree(c->d_T1); dfree(c->d_T2);
    dfree(c->d_Abar); dfree(c->d_BT); dfree(c->d_AbarT); dfree(c->d_B16); dfree(c->d_f16scale); dfree(c->d_f16tmp); dfree(c->d_vecs); dfree(c->d_scalars); dfree(c->d_yy);
    dfree(c->d_flag); dfree(c->d_partial); dfree(c->d_work); dfree(c->d_grad);
    dfree(c->p_Xt); dfree(c->p_vpart); dfree(c->p_mupart); dfree(c->p_Phi);
    if (c->gexec) hipGraphExecDestroy(c->gexec);
    if (c->graph) hipGraphDestroy(c->graph);
    dfree(c->d_opt); dfree(c->d_tctr); dfree(c->d_hist);
    for (hipEvent_t e : c->pool) hipEventDestroy(e);
    if (c->copy_st) { hipStreamSynchronize(c->copy_st); hipStreamDestroy(c->copy_st); }
    if (c->comm && rccl().ok()) rccl().destroy(c->comm);
    if (c->ev_factor) hipEventDestroy(c->ev_factor);
    if (c->ev_fence) hipEventDestroy(c->ev_fence);
    if (c->h_pin) hipHostFree(c->h_pin);
    if (c->own_stream && c->st) hipStreamDestroy(c->st);
    delete c;
}

extern "C" const char* scfgp_last_error(const scfgp_ctx* c) { return c ? c->err.c_str() : "null context"; }

extern "C" int scfgp_set_params(scfgp_ctx* c, const double* params, int P) {
    if (!c || !params || P != c->g.P) { if (c) c->err = "set_params: wrong parameter count"; return SCFGP_EARG; }
    HIPCHK(c, hipSetDevice(c->device));
    c->h_params.assign(params, params + P);
    HIPCHK(c, hipMemcpyAsync(c->d_params, c->h_params.data(), sizeof(double) * P, hipMemcpyHostToDevice, c->st));
    unpack_params(c->g, c->d_params, c->d_F, c->d_Fall, c->d_Lall, c->d_Rall, c->d_sc, c->st);
    HIPCHK(c, hipGetLastError());
    c->have_params = true; c->cond_valid = false;
    return SCFGP_OK;
}

extern "C" int scfgp_get_params(scfgp_ctx* c, double* params, int P) {
    if (!c || !params || P != c->g.P || !c->have_params) return SCFGP_EARG;
    memcpy(params, c->h_params.data(), sizeof(double) * P);
    return SCFGP_OK;
}

// working set := rows of the store selected by idx (NULL = all rows, in order)
static int load_working_set(scfgp_ctx* c, const int64_t* d_idx, int64_t n, int64_t n_global) {
    if (int rc = ensure_rows(c, n)) return rc;
    c->Nglobal = n_global > 0 ? n_global : n;
    pack_data(c->g, c->d_Xraw, c->d_yraw, d_idx, c->d_Xt, c->d_y, c->st);
    sum_squares(c->d_y, c->g.Np, c->d_yy, 0, c->d_partial, c->st);
    HIPCHK(c, hipGetLastError());
    c->work_full = d_idx == nullptr;
    c->have_data = true; c->stage = 0; c->cond_valid = false; c->esc_denied = 0;
    return SCFGP_OK;
}

extern "C" int scfgp_set_data(scfgp_ctx* c, const double* X, const double* y, int64_t N, int64_t n_global) {
    if (!c || !X || !y || N < 1) { if (c) c->err = "set_data: bad arguments"; return SCFGP_EARG; }
    HIPCHK(c, hipSetDevice(c->device));
    if (N > c->store_cap) {
        HIPCHK(c, hipStreamSynchronize(c->st));
        dfree(c->d_Xraw); dfree(c->d_yraw); c->store_cap = 0;
        if (int rc = dmalloc(c, &c->d_Xraw, sizeof(double) * N * c->g.D)) return rc;
        if (int rc = dmalloc(c, &c->d_yraw, sizeof(double) * N)) return rc;
        c->store_cap = N;
    }
    c->Nstore = N; c->Nglobal_full = n_global > 0 ? n_global : N;
    HIPCHK(c, hipMemcpyAsync(c->d_Xraw, X, sizeof(double) * N * c->g.D, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemcpyAsync(c->d_yraw, y, sizeof(double) * N, hipMemcpyHostToDevice, c->st));
    if (int rc = load_working_set(c, nullptr, N, n_global)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->st));            // the caller may reuse X, y as soon as we return
    return SCFGP_OK;
}

// ----------------------------------------------------------------------------------------------
// staged evaluation
// ----------------------------------------------------------------------------------------------
// summed exchange buffer [packed | vec | scalars] -> full symmetric matrix + vector for the K x K stage
static void unpack_exchange(scfgp_ctx* c, const double* xp, double* x) {
    const Geom& g = c->g;
    unpack_tri_tiles(xp, g.Kp / g.tile, g.tile, x, g.Kp, c->st);
    hipMemcpyAsync(x + (int64_t)g.Kp * g.Kp, xp + c->n_pk, sizeof(double) * (g.Kp + 8), hipMemcpyDeviceToDevice, c->st);
}

template <typename T> struct Impl {
    typedef SweepKernels<T> SK;
    // out = [packed lower tiles of M^T diag(w) M | M^T side (Kp)],  M = Phi (pass 1) or V = Phi B (pass 2)
    static void gram_to(scfgp_ctx* c, const T* Mx, const double* w, const double* side, double* out, const char* name) {
        const Geom& g = c->g;
        const int nts = g.Kp / g.tile, ntiles = nts * (nts + 1) / 2;
        if constexpr (sizeof(T) == 4) {
            if (c->f16_gram()) {
                // f16x3 mode: a split pass over the operand (which also leaves the side vector's block partials), then the fp16 Gram
                // whose row chunks are the "splits" of the shared reduction
                const int nch = F16x3Kernels::gram_chunks(g, c->f16_chunk());
                { ProfScope ps(c, w ? "split_v" : "split_phi");
                  // V's own planes were written by the epilogue of V = Phi B (pass2)
                  if (w) F16x3Kernels::split_v(g, (const float*)Mx, w, side, nullptr, c->d_qV16g, c->d_f16side, c->d_f16tmp, c->st);
                  else F16x3Kernels::split_phi(g, (const float*)Mx, side, c->d_sc, c->d_Phi16, c->d_f16side, c->d_f16tmp, c->st); }
                { ProfScope ps(c, name);
                  F16x3Kernels::gram(g, w ? c->d_V16g : c->d_Phi16, w ? c->d_qV16g : c->d_Phi16, c->d_f16tmp + 4, c->f16_chunk(), c->d_slabs, c->st); }
                { ProfScope ps(c, "reduce_tiles"); reduce_tri_tiles(c->d_slabs, nch, nts, g.tile, out, c->st);
                  reduce_side(c->d_f16side, F16x3Kernels::side_blocks(g), g.Kp, g.gfull * g.tile + g.gstrip * 64, out + c->n_pk, c->st); }
                return;
            }
        }
        const int gs = c->splits.nsplit;
        double* sidepart = c->d_slabs + (size_t)gs * ntiles * g.tile * g.tile;
        { ProfScope ps(c, name);
          SK::gram(g, Mx, w, side, c->splits, c->dtype == SCFGP_F32 ? c->gram_chunk : 0, c->d_slabs, sidepart, c->d_flag + 8, c->d_ws2, c->st); }
        { ProfScope ps(c, "reduce_tiles"); reduce_tri_tiles(c->d_slabs, gs, nts, g.tile, out, c->st);
          reduce_side(sidepart, gs, g.Kp, g.gfull * g.tile + g.gstrip * 64, out + c->n_pk, c->st); }
    }
    static int pass1(scfgp_ctx* c) {
        const Geom& g = c->g;
        if (!c->in_train) HIPCHK(c, hipMemsetAsync(c->d_flag, 0, sizeof(int) * 4, c->st));
        const bool use64 = sizeof(T) == 4 && c->use64();
        c->last_cform = c->want_cform();
        if ((use64 && c->phi64_cap < g.Np) || (c->last_cform && c->c_cap < g.Np)) {
            c->err = "pass1: auxiliary row buffer missing"; return SCFGP_EARG;
        }
        const Projection proj{c->d_Fall, c->d_Lall, c->d_Rall, c->d_Tt};
        if constexpr (sizeof(T) == 4) {
            if (use64) {                                       // escalated pass 1: fp64 features, fp64 Gram and Phi^T y
                typedef SweepKernels<double> SK64;
                const int nts = g.Kp / g.tile, ntiles = nts * (nts + 1) / 2, gs = c->splits64.nsplit;
                double* sidepart = c->d_slabs + (size_t)gs * ntiles * g.tile * g.tile;
                { ProfScope ps(c, "featuremap64"); SK64::featuremap(g, c->d_Xt, proj, c->d_sc, (double*)c->d_Phi64, c->st); }
                { ProfScope ps(c, "gram64");
                  SK64::gram(g, (const double*)c->d_Phi64, nullptr, c->d_y, c->splits64, 0, c->d_slabs, sidepart, c->d_flag + 8, nullptr, c->st); }
                { ProfScope ps(c, "reduce_tiles"); reduce_tri_tiles(c->d_slabs, gs, nts, g.tile, c->d_xp1, c->st);
                  reduce_side(sidepart, gs, g.Kp, g.gfull * g.tile + g.gstrip * 64, c->d_xp1 + c->n_pk, c->st); }
            }
        }
        { ProfScope ps(c, "featuremap"); SK::featuremap(g, c->d_Xt, proj, c->d_sc, (T*)c->d_Phi, c->st); }
        if constexpr (sizeof(T) == 4) {
            // f16x3 mode: the apply tiles' packed pairs; with the fp16 Gram they come out of gram_to's split pass instead
            if (c->f16_on() && (use64 || !c->f16_gram())) {
                ProfScope ps(c, "split_phi");
                F16x3Kernels::split_phi(g, (const float*)c->d_Phi, c->d_y, c->d_sc, c->d_Phi16, c->d_f16side, c->d_f16tmp, c->st);
            }
        }
        if (!use64) gram_to(c, (const T*)c->d_Phi, nullptr, c->d_y, c->d_xp1, "gram");
        c->last_used64 = use64 || sizeof(T) == 8; c->last_level = sizeof(T) == 8 ? 0 : c->level();
        HIPCHK(c, hipMemcpyAsync(c->xs1(), c->d_yy, sizeof(double), hipMemcpyDeviceToDevice, c->st));
        write_status(c->xs1(), c->last_level >= 1 ? 1.0 : 0.0, c->cap() < 1 ? 1.0 : 0.0, c->cap() < 2 ? 1.0 : 0.0, 0.0, c->st);
        HIPCHK(c, hipGetLastError());
        return SCFGP_OK;
    }
    static int factor(scfgp_ctx* c) {
        const Geom& g = c->g;
        { ProfScope ps(c, "kstage_factor");
          hipMemcpyAsync(c->d_x1 + (int64_t)g.Kp * g.Kp, c->d_xp1 + c->n_pk, sizeof(double) * (g.Kp + 8), hipMemcpyDeviceToDevice, c->st);
          kstage_factor(c->kstage(), c->d_xp1, c->d_sc, c->st); }
        if (c->last_cform) {                                   // factor form: the typed operands are Li (in B's place) and Li^T (scratch)
            SK::convert(c->d_Li, (T*)c->d_BT, g.K, g.Kp, c->st);
            SK::convert_transposed(c->d_Li, (T*)c->d_AbarT, g.K, g.Kp, c->st);
        } else {
            SK::convert(c->d_B, (T*)c->d_BT, g.K, g.Kp, c->st);
            if (c->f16_on()) {
                F16x3Kernels::split_operand(g, c->d_B, c->d_B16, c->d_f16scale, c->d_partial, c->d_sc, c->st);
                F16x3Kernels::v_bound(g, c->d_B, c->d_sc, c->d_f16tmp, c->st);
            }
        }
        HIPCHK(c, hipGetLastError());
        return SCFGP_OK;
    }
    static int pass2(scfgp_ctx* c, int want_grad) {
        const Geom& g = c->g;
        if (c->last_cform) {
            // triangular products: 128-wide LDS-DMA tiles wherever the square products use LDS-DMA.  With the column tile rotated
            // by the row block (apply.hip: the shader engines' round-robin) they beat the loader-staged tiles that pair a long
            // and a short column tile per workgroup: C3 34.4 / 33.5 ms against 40.0 / 34.9 (profiles/r05_tuning.md)
            const int dma = c->dma() ? 1 : 0;
            { ProfScope ps(c, "apply_c");
              SK::apply_c(g, (const T*)c->d_Phi, (const T*)c->d_AbarT, (const T*)c->d_BT, (T*)c->d_C, c->d_vpart, c->alpha(), c->beta(), c->d_mu, c->st, dma); }
            { ProfScope ps(c, "apply_vc");
              SK::apply_vc(g, (const T*)c->d_C, (const T*)c->d_BT, (const T*)c->d_AbarT, (T*)c->d_V, c->st, dma); }
        } else {
            ProfScope ps(c, "apply_v");
            F16Operands f16 = c->f16ops();
            if (want_grad && c->f16_gram()) f16.V16 = c->d_V16g;      // the weighted Gram's operand, from this product's epilogue
            SK::apply_v(g, (const T*)c->d_Phi, (const T*)c->d_BT, (T*)c->d_V, c->d_vpart, c->alpha(), c->d_mu, c->st, c->dma(), c->f16_on() ? &f16 : nullptr);
        }
        const int nb = (int)std::min<int64_t>(g.Np / 4, 2048);
        { ProfScope ps(c, "rowstats");
          SK::rowstats(g, c->d_mu, c->d_vpart, c->d_y, c->d_sc, c->d_p, c->d_q, c->d_partial, nb, c->st);
          reduce_scalars(c->d_partial, nb, 4, c->xs2(), 0, c->st);       // T2, kbar, sum q v, sum p mu
          write_status(c->xs2(), 0.0, 0.0, 0.0, 0.0, c->st); }
        if (want_grad) {
            // factor form: C^T diag(q) C and C^T p (the K x K stage turns them into B W B and u); else V^T diag(q) V = B W B, V^T p = u
            gram_to(c, c->last_cform ? (const T*)c->d_C : (const T*)c->d_V, c->d_q, c->d_p, c->d_xp2, "gram_w");
        }
        HIPCHK(c, hipGetLastError());
        return SCFGP_OK;
    }
    static int adjoint(scfgp_ctx* c) {
        const Geom& g = c->g;
        { ProfScope ps(c, "kstage_adjoint"); unpack_exchange(c, c->d_xp2, c->d_x2);
          if (c->last_cform) kstage_adjoint_factor_form(c->kstage(), c->d_x2, c->d_Abar, c->d_sc, c->st);
          else kstage_adjoint(c->kstage(), c->d_x2, c->d_Abar, c->d_sc, c->st);
          // the K x K part of bbar from the summed G (exchange buffer 1 is intact until the next pass 1) and Abar
          kstage_bbar(c->kstage(), c->d_xp1, c->d_Abar, c->d_partial, c->st); }
        SK::convert(c->d_Abar, (T*)c->d_AbarT, g.K, g.Kp, c->st);
        if (c->f16_on()) F16x3Kernels::split_operand(g, c->d_Abar, c->d_B16, c->d_f16scale, c->d_partial, c->d_sc, c->st);
        HIPCHK(c, hipGetLastError());
        return SCFGP_OK;
    }
    static int pass3(scfgp_ctx* c) {
        const Geom& g = c->g;
        { ProfScope ps(c, "apply_phibar");
          const F16Operands f16 = c->f16ops();
          SK::apply_phibar(g, (const T*)c->d_Phi, (const T*)c->d_AbarT, (T*)c->d_V, c->d_p, c->d_q, c->d_y, c->alpha(), c->ut(), c->st, c->dma(),
                           c->f16_on() ? &f16 : nullptr);
          write_status(c->x3_scalars(), 0.0, 0.0, 0.0, 0.0, c->st); }
        const int ntm = c->Dpp / XT, ntn = g.Jp / XT;
        const int64_t chunk = c->dtype == SCFGP_F32 ? c->gram_chunk : 0;
        c->last_lrb = c->want_lrb();
        if (c->last_lrb) {
            // rank-S form of the backward projection (F = l_F r_F^T, SCFGP.py:83,100): Zbar once, in place over Phibar's cosine half;
            // T~^T Zbar (T~ = [X l_F | 1] is resident from the forward projection); U = Zbar_L + Zbar_M r_F into Phibar's dead
            // sine half; X~^T U.  2 N (S+1) J + 2 N J S + 2 N (D+1) S flops instead of 2 N (D+1) J.
            ProfScope ps(c, "xtz");
            T* Zb = (T*)c->d_V; T* U = (T*)c->d_V + g.Jp;
            SK::zbar_inplace(g, (const T*)c->d_Phi, Zb, c->st);
            SK::rsel(g, c->d_params, (T*)c->d_AbarT, c->st);                       // Abar's typed copy is dead after the product above
            SK::apply_plain(g, Zb, (const T*)c->d_AbarT, U, g.J, g.S, c->st);
            const int nt1 = (c->Spp / XT) * ntn, xs1 = xtz_split(nt1, g.Np);
            SK::tn_plain(c->d_Tt, g.Sp, Zb, g.Kp, g.J, g.Np, xs1, chunk, c->d_slabs, c->st);
            reduce_full_tiles(c->d_slabs, xs1, c->Spp / XT, ntn, c->d_x3, g.Jp, c->st);
            const int nt2 = ntm * (c->Sq / XT), xs2 = xtz_split(nt2, g.Np);
            SK::tn_plain(c->d_Xt, g.Dp, U, g.Kp, g.S, g.Np, xs2, chunk, c->d_slabs, c->st);
            reduce_full_tiles(c->d_slabs, xs2, ntm, c->Sq / XT, c->x3_xu(), c->Sq, c->st);
            HIPCHK(c, hipGetLastError());
            return SCFGP_OK;
        }
        const int xs = c->xtz_nsplit > 0 ? (int)std::min<int64_t>(c->xtz_nsplit, g.Np / 64) : xtz_split(ntm * ntn, g.Np);
        { ProfScope ps(c, "xtz");
          SK::xtz(g, c->d_Xt, (const T*)c->d_Phi, (const T*)c->d_V, xs, chunk, c->d_slabs, c->st);
          reduce_full_tiles(c->d_slabs, xs, ntm, ntn, c->d_x3, g.Jp, c->st); }
        HIPCHK(c, hipGetLastError());
        return SCFGP_OK;
    }
    static int predict_chunk(scfgp_ctx* c, const Geom& g, const T* Bt, double* mu, double* sd) {
        SK::featuremap(g, c->p_Xt, Projection{c->d_Fall, c->d_Lall, c->d_Rall, c->p_Tt}, c->d_sc, (T*)c->p_Phi, c->st);
        SK::apply_predict(g, (const T*)c->p_Phi, Bt, c->p_vpart, c->alpha_pred(), c->p_mupart, c->st);           // Bt = Li^T
        SK::rowpredict(g, c->p_mupart, c->p_vpart, c->d_sc, mu, sd, c->st);
        HIPCHK(c, hipGetLastError());
        return SCFGP_OK;
    }
};

#define DISPATCH(c, fn, ...) ((c)->dtype == SCFGP_F32 ? Impl<float>::fn(__VA_ARGS__) : Impl<double>::fn(__VA_ARGS__))

static int ready(scfgp_ctx* c) {
    if (!c) return SCFGP_EARG;
    if (!c->have_params || !c->have_data) { c->err = "parameters and data must be set first"; return SCFGP_EARG; }
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return SCFGP_EHIP; }
    return SCFGP_OK;
}

// scfgp_eval_rows leaves a gathered minibatch as the working set; every entry point that evaluates "the resident
// rows" (scfgp_eval with X == NULL, the staged path, scfgp_train) first brings all rows of the data set back,
// with the n_global it was uploaded with (the reference refits on all N rows after minibatch training, SCFGP.py:265)
static int restore_full_set(scfgp_ctx* c) {
    if (!c->have_data || c->work_full) return SCFGP_OK;
    return load_working_set(c, nullptr, c->Nstore, c->Nglobal_full);
}

// with a communicator attached (scfgp_comm_init) every sweep ends in its sum over the ranks, enqueued on the context's own
// stream right behind the sweep: no fences, no host in between
static int comm_sum(scfgp_ctx* c, int stage, const char* name) {
    if (!c->comm) return SCFGP_OK;
    void* p = nullptr; int64_t n = 0;
    if (int rc = scfgp_exchange(c, stage, &p, &n)) return rc;
    ProfScope ps(c, name);
    const ncclResult_t r = rccl().all_reduce(p, p, (size_t)n, ncclFloat64, ncclSum, c->comm, c->st);
    if (r != ncclSuccess) { c->err = std::string("ncclAllReduce: ") + rccl().err_string(r); return SCFGP_EHIP; }
    return SCFGP_OK;
}

// test-only fault injection (option "test_fail_stage"): the next call of sweep `stage` fails before it enqueues anything
static int injected_failure(scfgp_ctx* c, int stage) {
    if (c->test_fail_stage != stage) return SCFGP_OK;
    c->test_fail_stage = 0;
    c->err = "pass" + std::to_string(stage) + ": injected failure (option test_fail_stage)";
    return SCFGP_EHIP;
}

// Ranks commit to ONE precision level (auto policy of fp32 mode).  A level is raised when the condition estimate asks for it --
// the estimate comes from the summed matrix, so every rank tries at the same evaluation -- but whether the row buffers of the
// level can be allocated is a rank's own affair.  The evaluation after such an attempt therefore carries, in the scalar tail of
// exchange 1 (common.h: XS_CAP1, XS_CAP2, XS_RAN1), how many ranks cannot reach level 1 / level 2 and how many formed their
// pass-1 Gram in fp64; once the sum over ranks is there every rank reads the same counts and
//   * takes the lowest attainable level as its own limit (a refusal anywhere is then recorded everywhere),
//   * continues if pass 1 of every rank already had the form of that level (levels 1 and 2 share pass 1),
//   * else asks for the stages again (SCFGP_REDO) -- all ranks do, the refusing one included --
// before pass 2, whose exchange would otherwise sum C^T diag(q) C on one rank with B W B on another.  One stream
// synchronisation, in that evaluation only.  Without peers the counts are the context's own and nothing changes.
static int settle_level(scfgp_ctx* c) {
    if (!c->unsettled || c->dtype != SCFGP_F32 || c->gram64 != 2) { c->unsettled = false; return SCFGP_OK; }
    double xs[8];
    HIPCHK(c, hipMemcpyAsync(xs, c->xs1(), sizeof(xs), hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    c->unsettled = false;
    const int agreed = xs[XS_CAP1] > 0.5 ? 0 : (xs[XS_CAP2] > 0.5 ? 1 : 2);
    if (agreed < c->cap()) {
        c->esc_denied = agreed + 1;
        c->err = "precision level " + std::to_string(agreed + 1) + " refused on another rank; every rank stays at level " + std::to_string(agreed) + " or below";
    }
    if (c->esc_level > agreed) {
        c->esc_level = agreed;
        if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
        if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
        c->warm = false;
    }
    if (agreed < 1 && xs[XS_RAN1] > 0.5) { c->stage = 0; return SCFGP_REDO; }   // some rank's G is an fp64 one: everybody forms it again in fp32
    c->last_cform = c->want_cform(); c->last_level = c->level();               // pass 2 in the form of the agreed level
    return SCFGP_OK;
}

extern "C" int scfgp_pass1(scfgp_ctx* c) {
    if (int rc = ready(c)) return rc;
    if (int rc = restore_full_set(c)) return rc;
    if (c->prof) { c->recs.clear(); c->pool_used = 0; }
    if (int rc = injected_failure(c, 1)) return rc;
    if (int rc = DISPATCH(c, pass1, c)) return rc;
    c->stage = 1;
    return comm_sum(c, 1, "exchange1");
}
extern "C" int scfgp_factor(scfgp_ctx* c) {
    if (int rc = ready(c)) return rc;
    if (c->stage != 1) { c->err = "factor: call pass1 first"; return SCFGP_EARG; }
    if (int rc = settle_level(c)) return rc;                     // SCFGP_REDO: start again at scfgp_pass1
    if (int rc = DISPATCH(c, factor, c)) return rc;
    if (!g_scfgp_capturing) HIPCHK(c, hipEventRecord(c->ev_factor, c->st));
    c->stage = 2; return SCFGP_OK;
}
extern "C" int scfgp_pass2(scfgp_ctx* c, int want_grad) {
    if (int rc = ready(c)) return rc;
    if (c->stage != 2) { c->err = "pass2: call factor first"; return SCFGP_EARG; }
    if (int rc = injected_failure(c, 2)) return rc;
    if (int rc = DISPATCH(c, pass2, c, want_grad)) return rc;
    c->last_want_grad = want_grad; c->stage = 3;
    return comm_sum(c, 2, "exchange2");
}
extern "C" int scfgp_adjoint(scfgp_ctx* c) {
    if (int rc = ready(c)) return rc;
    if (c->stage != 3 || !c->last_want_grad) { c->err = "adjoint: call pass2(want_grad=1) first"; return SCFGP_EARG; }
    if (int rc = DISPATCH(c, adjoint, c)) return rc;
    c->stage = 4; return SCFGP_OK;
}
extern "C" int scfgp_pass3(scfgp_ctx* c) {
    if (int rc = ready(c)) return rc;
    if (c->stage != 4) { c->err = "pass3: call adjoint first"; return SCFGP_EARG; }
    if (int rc = injected_failure(c, 3)) return rc;
    if (int rc = DISPATCH(c, pass3, c)) return rc;
    c->stage = 5;
    return comm_sum(c, 3, "exchange3");
}

// A rank that cannot compute sweep `stage` of the evaluation in progress still owes its peers the sum that ends the sweep:
// this marks exchange buffer `stage` as failed (XS_FAIL; its other contents are whatever they were) and, with a communicator
// attached, enqueues the all-reduce.  Called for every remaining exchange of the evaluation (want_grad sizes exchange 2 and
// decides whether there is a third), it lets the peers run to their scfgp_finish, where the summed XS_FAIL makes all of them
// return SCFGP_EPEER -- nobody waits in a collective for a rank that has left.  A sticky HIP error
// (a faulted kernel) cannot be helped this way: the calls below fail too.
extern "C" int scfgp_fail_stage(scfgp_ctx* c, int stage, int want_grad) {
    if (!c || stage < 1 || stage > 3 || (stage == 3 && !want_grad)) return SCFGP_EARG;
    const std::string keep = c->err;                             // the caller still wants the message of what failed
    if (hipSetDevice(c->device) != hipSuccess) return SCFGP_EHIP;
    c->last_want_grad = want_grad;
    double* xs = stage == 1 ? c->xs1() : (stage == 2 ? c->xs2() : c->x3_scalars());
    write_status(xs, 0.0, c->cap() < 1 ? 1.0 : 0.0, c->cap() < 2 ? 1.0 : 0.0, 1.0, c->st);
    const int rc = comm_sum(c, stage, stage == 1 ? "exchange1" : (stage == 2 ? "exchange2" : "exchange3"));
    c->stage = 0;
    if (rc == SCFGP_OK) c->err = keep;
    return rc;
}
// the exchanges of the evaluation in progress that have not been summed yet, all failed (native sums: scfgp_eval, scfgp_train)
static void fail_rest(scfgp_ctx* c, int want_grad) {
    const int st = c->stage, next = st < 1 ? 1 : (st < 3 ? 2 : (st < 5 ? 3 : 4));
    for (int s = next; s <= (want_grad ? 3 : 2); ++s)
        if (scfgp_fail_stage(c, s, want_grad) != SCFGP_OK) return;
}

extern "C" int scfgp_exchange(scfgp_ctx* c, int stage, void** dev_ptr, int64_t* count) {
    if (!c || !dev_ptr || !count) return SCFGP_EARG;
    if (stage == 1) { *dev_ptr = c->d_xp1; *count = c->n_xp; }
    else if (stage == 2) {
        if (c->last_want_grad) { *dev_ptr = c->d_xp2; *count = c->n_xp; }
        else { *dev_ptr = c->d_xp2 + c->n_pk + c->g.Kp; *count = 8; }    // forward only: just (T2, kbar)
    }
    else if (stage == 3) { *dev_ptr = c->d_x3; *count = c->n_x3; }
    else return SCFGP_EARG;
    return SCFGP_OK;
}

// Explicit ordering between the library's stream and the stream the host framework issues its collective on:
//   direction 0: `peer` waits for everything queued on the library's stream (call before the all-reduce)
//   direction 1: the library's stream waits for everything queued on `peer` (call after the all-reduce)
// A no-op when both are the same stream.  peer == NULL names the legacy default stream.
extern "C" int scfgp_stream_fence(scfgp_ctx* c, void* peer_stream, int direction) {
    if (!c || (direction != 0 && direction != 1)) return SCFGP_EARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t peer = (hipStream_t)peer_stream;
    if (peer == c->st) return SCFGP_OK;
    hipStream_t from = direction == 0 ? c->st : peer, to = direction == 0 ? peer : c->st;
    HIPCHK(c, hipEventRecord(c->ev_fence, from));
    HIPCHK(c, hipStreamWaitEvent(to, c->ev_fence, 0));
    return SCFGP_OK;
}

// alpha and Li are final once scfgp_factor has run; their copy to the host (K^2 doubles, 35.7 MB at
// K = 2112) goes on a second stream so it overlaps passes 2 and 3 instead of trailing the evaluation
extern "C" int scfgp_fetch_factors(scfgp_ctx* c, double* alpha, double* Li) {
    if (int rc = ready(c)) return rc;
    if (c->stage < 2) { c->err = "fetch_factors: call factor first"; return SCFGP_EARG; }
    const Geom& g = c->g;
    // The factors are final once the factor stage is done.  They cross PCIe on the copy stream into pinned memory
    // while the sweeps already queued behind the factor stage keep the GPU busy, and the host moves them on into the
    // caller's (pageable) arrays during those sweeps instead of after the last one.
    const size_t K = (size_t)g.K;
    if (!c->h_pin) HIPCHK(c, hipHostMalloc((void**)&c->h_pin, sizeof(double) * (K * K + K), hipHostMallocDefault));
    HIPCHK(c, hipStreamWaitEvent(c->copy_st, c->ev_factor, 0));
    if (alpha) HIPCHK(c, hipMemcpyAsync(c->h_pin + K * K, c->alpha(), sizeof(double) * K, hipMemcpyDeviceToHost, c->copy_st));
    if (Li) HIPCHK(c, hipMemcpy2DAsync(c->h_pin, sizeof(double) * K, c->d_Li, sizeof(double) * g.Kp, sizeof(double) * K, K,
                                       hipMemcpyDeviceToHost, c->copy_st));
    HIPCHK(c, hipStreamSynchronize(c->copy_st));
    if (alpha) memcpy(alpha, c->h_pin + K * K, sizeof(double) * K);
    if (Li) memcpy(Li, c->h_pin, sizeof(double) * K * K);
    return SCFGP_OK;
}

static void enqueue_epilogue(scfgp_ctx* c, int want_grad) {
    const Geom& g = c->g;
    ProfScope ps(c, "epilogue");
    if (want_grad && c->last_lrb)
        grad_epilogue(g, c->d_params, c->d_F, nullptr, 0, c->d_work, c->d_scalars, c->Nglobal, c->d_grad, c->st, c->d_x3, g.Jp, c->x3_xu(), c->Sq);
    else
        grad_epilogue(g, c->d_params, c->d_F, c->d_x3, g.Jp, c->d_work, c->d_scalars, c->Nglobal, want_grad ? c->d_grad : nullptr, c->st);
    finalize_cost(g, c->d_sc, c->d_scalars, c->xs1(), c->xs2(), c->Nglobal, c->d_grad, want_grad, c->st);
}

// Automatic precision level for the NEXT evaluation from the condition estimate of the one just finished (and, when that
// one ran below the level its own estimate asks for, SCFGP_REDO).  A Cholesky that failed on the fp32 Gram is retried on
// the fp64 one before it is reported.
static int update_level(scfgp_ctx* c, bool notpd, bool want_grad, bool may_redo = true) {
    if (c->dtype != SCFGP_F32 || c->gram64 != 2) return SCFGP_OK;
    const double est = c->cond_est();
    int want = notpd || !std::isfinite(est) ? 2 : c->want_level(est, 1.0);
    const int top = c->esc_denied > 0 ? std::min(want, c->esc_denied - 1) : want;     // levels refused before are not tried again
    if (top > c->esc_level) {
        // the buffers of the higher level first; the level is committed only when they exist.  If they cannot be allocated
        // (~17 GB at the headline shape for level 1, N Kp 4 bytes more for level 2) the context stays usable at the level below:
        // scfgp_get_condition keeps reporting the estimate and the predicted error, last_error says what was refused.  Row
        // shards: every rank is at this point in the same evaluation (the estimate is the summed matrix's), and the next
        // exchange 1 tells them all what each could allocate (settle_level) -- until then the level is `unsettled`.
        const int old = c->esc_level;
        c->unsettled = true;
        for (int lvl = top; lvl > old; --lvl) {                  // level 2 refused: level 1 may still fit
            c->esc_level = lvl;
            if (!(c->test_deny_level > 0 && lvl >= c->test_deny_level) && ensure_aux_rows(c) == SCFGP_OK) break;
            if (c->test_deny_level > 0 && lvl >= c->test_deny_level) c->err = "refused by option test_deny_level";
            (void)hipGetLastError();
            c->esc_level = lvl - 1; c->esc_denied = lvl;
            c->err = "precision level " + std::to_string(lvl) + " refused (auxiliary row buffers could not be allocated); staying at level " +
                     std::to_string(lvl - 1) + ": " + c->err;
        }
    }
    // Whether the evaluation is repeated depends on `top` -- the demand of the summed estimate under the limit the ranks last agreed
    // on, the same number on every rank -- and NOT on what this rank could allocate just now: a rank whose buffers were refused
    // repeats with the others (at its lower level) and tells them in the next exchange 1 (settle_level); deciding from its own
    // outcome it would return while they wait for it in that exchange.
    want = top;
    // level 2 only serves the gradient: a forward-only evaluation at level 1 is final
    if (may_redo && (want_grad ? want : std::min(want, 1)) > c->last_level) {
        if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
        if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
        c->warm = false;
        if (int rc = ensure_aux_rows(c)) return rc;
        if (notpd) HIPCHK(c, hipMemsetAsync(c->d_flag, 0, sizeof(int) * 4, c->st));
        return SCFGP_REDO;
    }
    if ((c->use64() && c->phi64_cap < c->g.Np) || (c->want_cform() && c->c_cap < c->g.Np)) {   // level raised without a repeat (scfgp_train, forward-only)
        if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
        if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
        c->warm = false;
        if (int rc = ensure_aux_rows(c)) return rc;
    }
    if (!notpd) {
        const int keep = c->want_level(est, 0.25);             // hysteresis: drop only well below the threshold
        if (keep < c->esc_level) {
            c->esc_level = keep;
            if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
            if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
            c->warm = false;
        }
    }
    return SCFGP_OK;
}

extern "C" int scfgp_finish(scfgp_ctx* c, int want_grad, double* cost, double* grad, double* alpha, double* Li) {
    if (int rc = ready(c)) return rc;
    if ((want_grad && c->stage != 5) || (!want_grad && c->stage != 3)) { c->err = "finish: evaluation incomplete"; return SCFGP_EARG; }
    const Geom& g = c->g;
    enqueue_epilogue(c, want_grad);
    double h_cost = 0, h_fail[3] = {0, 0, 0}; int h_flag[4] = {0, 0, 0, 0};
    {
        ProfScope ps(c, "d2h");
        HIPCHK(c, hipMemcpyAsync(&h_fail[0], c->xs1() + XS_FAIL, sizeof(double), hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipMemcpyAsync(&h_fail[1], c->xs2() + XS_FAIL, sizeof(double), hipMemcpyDeviceToHost, c->st));
        if (want_grad) HIPCHK(c, hipMemcpyAsync(&h_fail[2], c->x3_scalars() + XS_FAIL, sizeof(double), hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipMemcpyAsync(&h_cost, c->d_scalars + R_COST, sizeof(double), hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipMemcpyAsync(c->cond, c->d_scalars + R_LMIN2, sizeof(double) * 3, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipMemcpyAsync(h_flag, c->d_flag, sizeof(int) * 4, hipMemcpyDeviceToHost, c->st));
        if (want_grad && grad) HIPCHK(c, hipMemcpyAsync(grad, c->d_grad, sizeof(double) * g.P, hipMemcpyDeviceToHost, c->st));
        if (alpha) HIPCHK(c, hipMemcpyAsync(alpha, c->alpha(), sizeof(double) * g.K, hipMemcpyDeviceToHost, c->st));
        if (Li) HIPCHK(c, hipMemcpy2DAsync(Li, sizeof(double) * g.K, c->d_Li, sizeof(double) * g.Kp, sizeof(double) * g.K, g.K,
                                           hipMemcpyDeviceToHost, c->st));
    }
    HIPCHK(c, hipStreamSynchronize(c->st));
    HIPCHK(c, hipGetLastError());
    c->stage = 0; c->warm = true;
    if (cost) *cost = h_cost;
    if (h_fail[0] + h_fail[1] + h_fail[2] > 0.5) {              // summed over the ranks: the same verdict everywhere, no level decision from it
        c->cond_valid = false;
        c->err = "a rank failed in this evaluation (exchange 1 / 2 / 3: " + std::to_string((int)h_fail[0]) + " / " + std::to_string((int)h_fail[1]) +
                 " / " + std::to_string((int)h_fail[2]) + " ranks): its results are not valid on any rank";
        return SCFGP_EPEER;
    }
    if (int rc = update_level(c, h_flag[0] != 0, want_grad != 0)) return rc;    // SCFGP_REDO: repeat the evaluation at the higher level
    c->cond_valid = true;
    if (h_flag[0]) { c->err = "Phi^T Phi + (e^{2a}+1e-6) I is not positive definite"; return SCFGP_ENOTPD; }
    if (!std::isfinite(h_cost)) { c->err = "non-finite cost"; return SCFGP_ENONFINITE; }
    return SCFGP_OK;
}

static int pass1_current(scfgp_ctx* c) {                  // pass 1 on the working set as it stands (full or gathered)
    if (int rc = ready(c)) return rc;
    if (c->prof) { c->recs.clear(); c->pool_used = 0; }
    if (int rc = injected_failure(c, 1)) return rc;
    if (int rc = DISPATCH(c, pass1, c)) return rc;
    c->stage = 1;
    return comm_sum(c, 1, "exchange1");
}
static int run_eval(scfgp_ctx* c, int want_grad, double* cost, double* grad, double* alpha, double* Li, bool subset = false) {
    int rc = SCFGP_OK;
    // with the sums inside the library a rank that fails still posts the remaining all-reduces of the evaluation (fail_rest)
    const auto leave = [&](int code) { if (code < 0 && c->comm) fail_rest(c, want_grad); return code; };
    for (int attempt = 0; attempt < 5; ++attempt) {              // at most two escalations (levels 0 -> 1 -> 2) and their settling
        if ((rc = subset ? pass1_current(c) : scfgp_pass1(c))) return leave(rc);
        if ((rc = scfgp_factor(c)) == SCFGP_REDO) continue;       // the ranks agreed on a lower level than some of them ran pass 1 at
        if (rc) return leave(rc);
        if ((rc = scfgp_pass2(c, want_grad))) return leave(rc);
        if (want_grad) {
            if ((rc = scfgp_adjoint(c))) return leave(rc);
            if ((rc = scfgp_pass3(c))) return leave(rc);
        }
        // everything is queued: the factor outputs now stream to the host beside passes 2 and 3
        if ((alpha || Li) && (rc = scfgp_fetch_factors(c, alpha, Li))) return rc;
        rc = scfgp_finish(c, want_grad, cost, grad, nullptr, nullptr);
        if (rc != SCFGP_REDO) return rc;
    }
    return rc;
}

extern "C" int scfgp_eval(scfgp_ctx* c, const double* X, const double* y, int64_t N, int want_grad,
                          double* cost, double* grad, double* alpha, double* Li) {
    if (!c) return SCFGP_EARG;
    int rc;
    if (X) { if ((rc = scfgp_set_data(c, X, y, N, N))) return rc; }
    return run_eval(c, want_grad, cost, grad, alpha, Li);      // scfgp_pass1 brings all rows back after a row subset
}

extern "C" int scfgp_eval_rows(scfgp_ctx* c, const int64_t* idx, int64_t n, int want_grad,
                               double* cost, double* grad, double* alpha, double* Li) {
    if (!c || !idx || n < 1) { if (c) c->err = "eval_rows: bad arguments"; return SCFGP_EARG; }
    if (!c->have_data || c->Nstore < 1) { c->err = "eval_rows: no resident data set"; return SCFGP_EARG; }
    for (int64_t i = 0; i < n; ++i)
        if (idx[i] < 0 || idx[i] >= c->Nstore) { c->err = "eval_rows: index out of range"; return SCFGP_EARG; }
    HIPCHK(c, hipSetDevice(c->device));
    if (n > c->idx_cap) {
        HIPCHK(c, hipStreamSynchronize(c->st));
        dfree(c->d_idx);
        if (int rc = dmalloc(c, &c->d_idx, sizeof(int64_t) * n)) return rc;
        c->idx_cap = n;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_idx, idx, sizeof(int64_t) * n, hipMemcpyHostToDevice, c->st));
    if (int rc = load_working_set(c, c->d_idx, n, n)) return rc;       // per-batch N, as SCFGP.py:126,128
    return run_eval(c, want_grad, cost, grad, alpha, Li, true);
}

// raw_mode: apply the registered X scaler while packing; post: y-scaler backward transform of the outputs on
// the device and, with targets ys, the six validation metrics
static int predict_impl(scfgp_ctx* c, const double* Xs, int64_t T, const double* alpha, const double* Li,
                        double* mu, double* sd, int raw_mode, int post = 0, const double* ys = nullptr,
                        double* metrics = nullptr) {
    if (!c || !Xs || !alpha || !Li || !mu || !sd || T < 1) { if (c) c->err = "predict: bad arguments"; return SCFGP_EARG; }
    if (!c->have_params) { c->err = "predict: parameters not set"; return SCFGP_EARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const Geom& g0 = c->g;
    const int64_t Kp = g0.Kp;
    const size_t ts = c->tsize();
    int rc;
    if (!c->p_Xt) {
        if ((rc = dmalloc(c, &c->p_Xt, sizeof(double) * PRED_ROWS * g0.Dp))) return rc;
        if (g0.lowrank && (rc = dmalloc(c, &c->p_Tt, sizeof(double) * PRED_ROWS * g0.Sp))) return rc;
        if ((rc = dmalloc(c, &c->p_vpart, sizeof(double) * PRED_ROWS * (Kp / 64)))) return rc;
        if ((rc = dmalloc(c, &c->p_mupart, sizeof(double) * PRED_ROWS * (Kp / 64)))) return rc;
        if ((rc = dmalloc(c, &c->p_Phi, ts * PRED_ROWS * Kp))) return rc;
        HIPCHK(c, hipMemsetAsync(c->p_Phi, 0, ts * PRED_ROWS * Kp, c->st));
    }
    // Li (K x K host) -> T1 (Kp x Kp, identity padding); typed transposed copy -> AbarT scratch
    DevTmp raw, d_out, d_ys, d_part;                              // Li / two chunks of Xs | mu, sd of all T rows | targets, mean, metrics | chunk partials
    const int64_t rawstride = PRED_ROWS * g0.D;
    if ((rc = dmalloc(c, &raw.p, sizeof(double) * std::max<int64_t>((int64_t)g0.K * g0.K, 2 * rawstride)))) return rc;
    if ((rc = dmalloc(c, &d_out.p, sizeof(double) * 2 * T))) return rc;
    double* d_mu = d_out; double* d_sd = d_out + T;
    HIPCHK(c, hipMemcpyAsync(raw, Li, sizeof(double) * g0.K * g0.K, hipMemcpyHostToDevice, c->st));
    pad_square(raw, g0.K, g0.Kp, c->d_T1, c->st);
    HIPCHK(c, hipMemsetAsync(c->alpha_pred(), 0, sizeof(double) * Kp, c->st));
    HIPCHK(c, hipMemcpyAsync(c->alpha_pred(), alpha, sizeof(double) * g0.K, hipMemcpyHostToDevice, c->st));
    // the sweep operand is Li^T itself: sigma* needs rowsum((Phi* Li^T)^2) (SCFGP/SCFGP.py:144), a triangular product
    const void* Bt = c->d_AbarT;                                  // AbarT is scratch outside adjoint..pass3
    if (c->dtype == SCFGP_F32) SweepKernels<float>::convert_transposed(c->d_T1, (float*)c->d_AbarT, g0.K, g0.Kp, c->st);
    else SweepKernels<double>::convert_transposed(c->d_T1, (double*)c->d_AbarT, g0.K, g0.Kp, c->st);
    HIPCHK(c, hipStreamSynchronize(c->st));                     // raw is reused below
    const int nchunks = (int)((T + PRED_ROWS - 1) / PRED_ROWS);
    if (post && ys) {
        if ((rc = dmalloc(c, &d_ys.p, sizeof(double) * (T + 8)))) return rc;
        if ((rc = dmalloc(c, &d_part.p, sizeof(double) * 4 * YPOST_BLOCKS * nchunks))) return rc;
        HIPCHK(c, hipMemcpyAsync(d_ys, ys, sizeof(double) * T, hipMemcpyHostToDevice, c->st));
        ypost_mean(d_ys, T, d_ys + T, c->st);
    }
    // chunk i+1 is uploaded (pageable host memory: the call blocks the host while it stages) on the copy stream while the
    // kernels of chunk i run; raw holds two chunks, ev_up / ev_free order its two halves between the streams
    struct Events {
        hipEvent_t up[2] = {nullptr, nullptr}, fre[2] = {nullptr, nullptr};
        ~Events() { for (int i = 0; i < 2; ++i) { if (up[i]) (void)hipEventDestroy(up[i]); if (fre[i]) (void)hipEventDestroy(fre[i]); } }
    } ev;
    for (int i = 0; i < 2; ++i) {
        HIPCHK(c, hipEventCreateWithFlags(&ev.up[i], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&ev.fre[i], hipEventDisableTiming));
    }
    auto upload = [&](int64_t i) -> int {
        const int64_t t0 = i * PRED_ROWS, n = std::min<int64_t>(PRED_ROWS, T - t0);
        const int h = (int)(i & 1);
        if (i >= 2) HIPCHK(c, hipStreamWaitEvent(c->copy_st, ev.fre[h], 0));
        HIPCHK(c, hipMemcpyAsync(raw + h * rawstride, Xs + t0 * g0.D, sizeof(double) * n * g0.D, hipMemcpyHostToDevice, c->copy_st));
        HIPCHK(c, hipEventRecord(ev.up[h], c->copy_st));
        return SCFGP_OK;
    };
    if ((rc = upload(0))) return rc;
    for (int64_t i = 0; i < nchunks; ++i) {
        const int64_t t0 = i * PRED_ROWS;
        const int h = (int)(i & 1);
        Geom g = g0;
        g.N = std::min<int64_t>(PRED_ROWS, T - t0); g.Np = round_up(g.N, 256);
        HIPCHK(c, hipStreamWaitEvent(c->st, ev.up[h], 0));
        pack_data(g, raw + h * rawstride, nullptr, nullptr, c->p_Xt, nullptr, c->st, raw_mode ? c->xs_mode : 0, c->d_xscale);
        HIPCHK(c, hipEventRecord(ev.fre[h], c->st));
        rc = c->dtype == SCFGP_F32 ? Impl<float>::predict_chunk(c, g, (const float*)Bt, d_mu + t0, d_sd + t0)
                                   : Impl<double>::predict_chunk(c, g, (const double*)Bt, d_mu + t0, d_sd + t0);
        if (rc) return rc;
        if (post)
            ypost_chunk(d_mu + t0, d_sd + t0, d_ys.p ? d_ys + t0 : nullptr, g.N, c->ys_mode, c->d_yscale, d_ys.p ? d_ys + T : nullptr,
                        d_part.p ? d_part + 4 * YPOST_BLOCKS * i : nullptr, c->st);
        if (i + 1 < nchunks && (rc = upload(i + 1))) return rc;
    }
    HIPCHK(c, hipMemcpyAsync(mu, d_mu, sizeof(double) * T, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipMemcpyAsync(sd, d_sd, sizeof(double) * T, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (post && ys) {
        ypost_metrics(d_part, YPOST_BLOCKS * nchunks, T, d_ys + T + 1, c->st);
        HIPCHK(c, hipMemcpyAsync(metrics, d_ys + T + 1, sizeof(double) * 6, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
    }
    HIPCHK(c, hipGetLastError());
    return SCFGP_OK;
}

extern "C" int scfgp_predict(scfgp_ctx* c, const double* Xs, int64_t T, const double* alpha, const double* Li,
                             double* mu, double* sd) {
    return predict_impl(c, Xs, T, alpha, Li, mu, sd, 0);
}

// SCFGP.predict feeds pred_func with X_scaler.forward_transform(Xs) (SCFGP/SCFGP.py:279); for large test
// sets that element-wise host pass dominates, so the same transform can run inside the packing kernel.
extern "C" int scfgp_set_x_scaler(scfgp_ctx* c, int mode, const double* mn, const double* mx, const double* boxcox,
                                  const double* mu, const double* sd) {
    if (!c || mode < 0 || mode > 5) return SCFGP_EARG;
    HIPCHK(c, hipSetDevice(c->device));
    const int D = c->g.D;
    std::vector<double> h(5 * D, 0.0);
    const double* src[5] = {mn, mx, boxcox, mu, sd};
    for (int k = 0; k < 5; ++k)
        if (src[k]) std::copy(src[k], src[k] + D, h.begin() + k * D);
    if (!c->d_xscale) { if (int rc = dmalloc(c, &c->d_xscale, sizeof(double) * 5 * D)) return rc; }
    HIPCHK(c, hipMemcpy(c->d_xscale, h.data(), sizeof(double) * 5 * D, hipMemcpyHostToDevice));
    c->xs_mode = mode;
    return SCFGP_OK;
}
extern "C" int scfgp_predict_raw(scfgp_ctx* c, const double* Xs, int64_t T, const double* alpha, const double* Li,
                                 double* mu, double* sd) {
    if (c && c->xs_mode && !c->d_xscale) { c->err = "predict_raw: no scaler set"; return SCFGP_EARG; }
    return predict_impl(c, Xs, T, alpha, Li, mu, sd, 1);
}

// The rest of SCFGP.predict (SCFGP/SCFGP.py:281-293): y_scaler.backward_transform of mu and of the mu +- std
// band, std_y = half the transformed band, and with targets the metrics MAE, NMAE, MSE, NMSE, MNLP, SCORE.
extern "C" int scfgp_set_y_scaler(scfgp_ctx* c, int mode, double mn, double mx, double boxcox, double mu, double sd) {
    if (!c || mode < 0 || mode > 5) return SCFGP_EARG;
    HIPCHK(c, hipSetDevice(c->device));
    const double h[5] = {mn, mx, boxcox, mu, sd};
    if (!c->d_yscale) { if (int rc = dmalloc(c, &c->d_yscale, sizeof(double) * 5)) return rc; }
    HIPCHK(c, hipMemcpy(c->d_yscale, h, sizeof(h), hipMemcpyHostToDevice));
    c->ys_mode = mode;
    return SCFGP_OK;
}
extern "C" int scfgp_predict_y(scfgp_ctx* c, const double* Xs, int64_t T, const double* alpha, const double* Li,
                               const double* ys, double* mu_y, double* std_y, double* metrics) {
    if (c && ((c->xs_mode && !c->d_xscale) || !c->d_yscale)) { c->err = "predict_y: scalers not set"; return SCFGP_EARG; }
    if (c && ys && !metrics) { c->err = "predict_y: targets given without a metrics buffer"; return SCFGP_EARG; }
    return predict_impl(c, Xs, T, alpha, Li, mu_y, std_y, 1, 1, ys, metrics);
}

// ----------------------------------------------------------------------------------------------
// on-device optimiser and the captured training iteration
// ----------------------------------------------------------------------------------------------
extern "C" int scfgp_opt_init(scfgp_ctx* c, int algo, const double* hyper, int nhyper, double momentum) {
    if (!c || algo < 0 || algo > 5 || !hyper || nhyper < 4) { if (c) c->err = "opt_init: bad arguments"; return SCFGP_EARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const int P = c->g.P;
    if (!c->d_opt) {
        if (int rc = dmalloc(c, &c->d_opt, sizeof(double) * 3 * P)) return rc;
        if (int rc = dmalloc(c, &c->d_tctr, sizeof(double) * 8)) return rc;
    }
    HIPCHK(c, hipMemsetAsync(c->d_opt, 0, sizeof(double) * 3 * P, c->st));
    HIPCHK(c, hipMemsetAsync(c->d_tctr, 0, sizeof(double) * 8, c->st));
    c->opt_algo = algo;
    c->opt_h = OptHyper{hyper[0], hyper[1], hyper[2], hyper[3], momentum};
    if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
    if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
    HIPCHK(c, hipStreamSynchronize(c->st));
    return SCFGP_OK;
}

// which: 0 = s1, 1 = s2, 2 = velocity (P doubles each), 3 = step counter t (1 double)
extern "C" int scfgp_opt_state(scfgp_ctx* c, int set, int which, double* buf) {
    if (!c || !buf || c->opt_algo < 0 || which < 0 || which > 3) return SCFGP_EARG;
    HIPCHK(c, hipSetDevice(c->device));
    double* dev = which == 3 ? c->d_tctr : c->d_opt + (int64_t)which * c->g.P;
    const size_t bytes = sizeof(double) * (which == 3 ? 1 : c->g.P);
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (set) HIPCHK(c, hipMemcpy(dev, buf, bytes, hipMemcpyHostToDevice));
    else HIPCHK(c, hipMemcpy(buf, dev, bytes, hipMemcpyDeviceToHost));
    return SCFGP_OK;
}

// One step of the device rule with a caller-supplied gradient (P doubles on the host): the parameter vector and the rule's
// state advance exactly as inside scfgp_train, without an evaluation -- for callers that form the gradient elsewhere and for
// known-answer tests of the rule itself (tests/golden/optimizer_kats.npz).
extern "C" int scfgp_opt_step(scfgp_ctx* c, const double* grad, int P) {
    if (!c || !grad || P != c->g.P) { if (c) c->err = "opt_step: wrong gradient length"; return SCFGP_EARG; }
    if (c->opt_algo < 0 || !c->have_params) { c->err = "opt_step: call scfgp_set_params and scfgp_opt_init first"; return SCFGP_EARG; }
    if (c->stage != 0) { c->err = "opt_step: a staged evaluation is in progress (finish it first: its later stages would run on other parameters)"; return SCFGP_EARG; }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->d_grad, grad, sizeof(double) * P, hipMemcpyHostToDevice, c->st));
    opt_update(c->opt_algo, c->opt_h, P, c->d_params, c->d_grad, c->d_opt, c->d_tctr, c->d_scalars, nullptr, 0, c->st);
    unpack_params(c->g, c->d_params, c->d_F, c->d_Fall, c->d_Lall, c->d_Rall, c->d_sc, c->st);
    HIPCHK(c, hipMemcpyAsync(c->h_params.data(), c->d_params, sizeof(double) * P, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    HIPCHK(c, hipGetLastError());
    c->cond_valid = false;
    return SCFGP_OK;
}

// One training iteration, enqueued.  With a communicator attached the three sums over ranks are part of it (ncclAllReduce on
// the context's stream right behind each sweep); every rank then applies the same deterministic rule to the same summed
// gradient, so the parameter vectors stay bit-equal without a broadcast.  c->stage follows the sweeps so that a failing rank
// can tell which exchanges it still owes (fail_rest).
static int enqueue_train_iter(scfgp_ctx* c) {
    int rc;
    c->stage = 0; c->last_want_grad = 1;
    if ((rc = injected_failure(c, 1))) return rc;
    if ((rc = DISPATCH(c, pass1, c))) return rc;
    c->stage = 1;
    if ((rc = comm_sum(c, 1, "exchange1"))) return rc;
    if ((rc = DISPATCH(c, factor, c))) return rc;
    c->stage = 2;
    if ((rc = injected_failure(c, 2))) return rc;
    if ((rc = DISPATCH(c, pass2, c, 1))) return rc;
    c->stage = 3;
    if ((rc = comm_sum(c, 2, "exchange2"))) return rc;
    if ((rc = DISPATCH(c, adjoint, c))) return rc;
    c->stage = 4;
    if ((rc = injected_failure(c, 3))) return rc;
    if ((rc = DISPATCH(c, pass3, c))) return rc;
    c->stage = 5;
    if ((rc = comm_sum(c, 3, "exchange3"))) return rc;
    enqueue_epilogue(c, 1);
    opt_update(c->opt_algo, c->opt_h, c->g.P, c->d_params, c->d_grad, c->d_opt, c->d_tctr, c->d_scalars, c->d_hist, c->hist_cap, c->st);
    unpack_params(c->g, c->d_params, c->d_F, c->d_Fall, c->d_Lall, c->d_Rall, c->d_sc, c->st);
    HIPCHK(c, hipGetLastError());
    return SCFGP_OK;
}

extern "C" int scfgp_train(scfgp_ctx* c, int n_iters, double* cost_hist, double* alpha, double* Li) {
    if (int rc = ready(c)) return rc;
    if (c->opt_algo < 0 || n_iters < 1) { c->err = "train: call scfgp_opt_init first"; return SCFGP_EARG; }
    if (int rc = restore_full_set(c)) return rc;
    if (!c->comm && c->Nglobal != c->g.N) {                      // a shard without its peers would train on its own rows' sums
        c->err = "train: the resident rows are a shard (n_global != N) but no communicator is attached (scfgp_comm_init); the "
                 "on-device loop needs the sums over ranks inside the library";
        return SCFGP_EARG;
    }
    const Geom& g = c->g;
    if (n_iters > c->hist_cap) {
        dfree(c->d_hist);
        if (int rc = dmalloc(c, &c->d_hist, sizeof(double) * n_iters)) return rc;
        c->hist_cap = n_iters;
        if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
        if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
    }
    // auto precision level, condition of the current parameters unknown (no evaluation since they or the rows were set):
    // one pass 1 + factor stage as a probe, so that the iterations of this call run at the level their first one needs
    // (row shards: the probe ends in exchange 1 like any pass 1, and a level raised or refused since the last sum over ranks
    // is settled here, before the iterations -- possibly one captured graph -- are enqueued)
    for (int round = 0; round < 5 && c->dtype == SCFGP_F32 && c->gram64 == 2 && (!c->cond_valid || c->unsettled); ++round) {
        int rc;
        c->stage = 0;
        if ((rc = injected_failure(c, 1)) || (rc = DISPATCH(c, pass1, c))) {
            if (c->comm) (void)scfgp_fail_stage(c, 1, 1);         // the peers' probes wait in exchange 1; they read the mark and stop there too
            return rc;
        }
        c->stage = 1;
        if ((rc = comm_sum(c, 1, "exchange1"))) return rc;
        if ((rc = settle_level(c)) == SCFGP_REDO) continue;
        if (rc) return rc;
        if ((rc = DISPATCH(c, factor, c))) return rc;
        int h_flag[4] = {0, 0, 0, 0}; double h_fail = 0;
        HIPCHK(c, hipMemcpyAsync(c->cond, c->d_scalars + R_LMIN2, sizeof(double) * 3, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipMemcpyAsync(h_flag, c->d_flag, sizeof(int) * 4, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipMemcpyAsync(&h_fail, c->xs1() + XS_FAIL, sizeof(double), hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
        c->stage = 0;
        if (h_fail > 0.5) { c->err = "train: a rank failed in the probe evaluation"; return SCFGP_EPEER; }
        if ((rc = update_level(c, h_flag[0] != 0, true, false))) return rc;
        c->cond_valid = true;
    }
    if (c->unsettled) { c->err = "train: the ranks' precision level did not settle"; return SCFGP_EHIP; }
    HIPCHK(c, hipMemsetAsync(c->d_flag, 0, sizeof(int) * 4, c->st));
    HIPCHK(c, hipMemsetAsync(c->d_tctr + 1, 0, sizeof(double), c->st));
    c->in_train = true;
    int rc = SCFGP_OK, done = 0;
    // the captured iteration: always without a communicator; with one only on request (option use_graph = 2) -- capturing
    // ncclAllReduce is RCCL's business and has not run here with more than one rank, eager launches queue far ahead of the GPU anyway
    const bool graph_ok = c->use_graph && !c->prof && (!c->comm || c->use_graph >= 2);
    if (graph_ok && !c->warm) {                                  // first touch of every kernel: eager
        rc = enqueue_train_iter(c); done = 1;
        if (rc == SCFGP_OK) c->warm = true;
    }
    if (rc == SCFGP_OK && graph_ok && done < n_iters) {
        if (!c->gexec || c->graph_N != g.N) {
            if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
            if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
            hipError_t e = hipStreamBeginCapture(c->st, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                g_scfgp_capturing = true;
                rc = enqueue_train_iter(c);
                g_scfgp_capturing = false;
                e = hipStreamEndCapture(c->st, &c->graph);
                if (rc == SCFGP_OK && e == hipSuccess) e = hipGraphInstantiate(&c->gexec, c->graph, nullptr, nullptr, 0);
            }
            if (e != hipSuccess || rc != SCFGP_OK) {                 // fall back to eager launches
                if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
                c->gexec = nullptr; (void)hipGetLastError(); rc = SCFGP_OK;
            } else c->graph_N = g.N;
        }
        if (c->gexec)
            for (; done < n_iters && rc == SCFGP_OK; ++done)
                if (hipGraphLaunch(c->gexec, c->st) != hipSuccess) { c->err = "hipGraphLaunch failed"; rc = SCFGP_EHIP; }
    }
    for (; done < n_iters && rc == SCFGP_OK; ++done) {
        if (c->prof) { c->recs.clear(); c->pool_used = 0; }      // timings describe the last eager iteration
        rc = enqueue_train_iter(c);
    }
    c->in_train = false;
    if (rc != SCFGP_OK) {
        // the peers are enqueueing all n_iters iterations: this rank owes them every remaining sum (XS_FAIL set), or they wait
        // in an all-reduce for good; they learn of it from the summed XS_FAIL when their call ends
        if (c->comm && rc < 0) {
            fail_rest(c, 1);
            for (; done < n_iters; ++done) { c->stage = 0; fail_rest(c, 1); }
        }
        c->stage = 0;
        return rc;
    }
    int h_flag[4] = {0, 0, 0, 0}; double h_fail[3] = {0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(&h_fail[0], c->xs1() + XS_FAIL, sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipMemcpyAsync(&h_fail[1], c->xs2() + XS_FAIL, sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipMemcpyAsync(&h_fail[2], c->x3_scalars() + XS_FAIL, sizeof(double), hipMemcpyDeviceToHost, c->st));
    if (cost_hist) HIPCHK(c, hipMemcpyAsync(cost_hist, c->d_hist, sizeof(double) * n_iters, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipMemcpyAsync(c->h_params.data(), c->d_params, sizeof(double) * g.P, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipMemcpyAsync(h_flag, c->d_flag, sizeof(int) * 4, hipMemcpyDeviceToHost, c->st));
    if (alpha) HIPCHK(c, hipMemcpyAsync(alpha, c->alpha(), sizeof(double) * g.K, hipMemcpyDeviceToHost, c->st));
    if (Li) HIPCHK(c, hipMemcpy2DAsync(Li, sizeof(double) * g.K, c->d_Li, sizeof(double) * g.Kp, sizeof(double) * g.K, g.K,
                                       hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipMemcpyAsync(c->cond, c->d_scalars + R_LMIN2, sizeof(double) * 3, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    HIPCHK(c, hipGetLastError());
    c->stage = 0;
    if (h_fail[0] + h_fail[1] + h_fail[2] > 0.5) {
        c->cond_valid = false;
        c->err = "train: a rank failed; parameters and optimiser state of this call are not valid on any rank";
        return SCFGP_EPEER;
    }
    // the precision level is fixed inside one call; the last iteration's condition estimate sets it for the next call
    if (int rc = update_level(c, h_flag[0] != 0, true, false)) return rc;
    c->cond_valid = true;
    if (h_flag[0]) { c->err = "Phi^T Phi + (e^{2a}+1e-6) I is not positive definite"; return SCFGP_ENOTPD; }
    if (cost_hist) for (int i = 0; i < n_iters; ++i) if (!std::isfinite(cost_hist[i])) { c->err = "non-finite cost"; return SCFGP_ENONFINITE; }
    return SCFGP_OK;
}

// ----------------------------------------------------------------------------------------------
// row-sharded evaluation with the sums inside the library (RCCL all-reduce over xGMI)
// ----------------------------------------------------------------------------------------------
extern "C" int scfgp_comm_unique_id(void* id128) {
    if (!id128) return SCFGP_EARG;
    if (!rccl().ok()) return SCFGP_EHIP;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    return rccl().get_id((ncclUniqueId*)id128) == ncclSuccess ? SCFGP_OK : SCFGP_EHIP;
}
extern "C" int scfgp_comm_init(scfgp_ctx* c, int nranks, int rank, const void* id128) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) { if (c) c->err = "comm_init: bad arguments"; return SCFGP_EARG; }
    if (c->comm) { c->err = "comm_init: this context already has a communicator"; return SCFGP_EARG; }
    if (!rccl().ok()) { c->err = "comm_init: librccl.so could not be loaded"; return SCFGP_EHIP; }
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId id; memcpy(&id, id128, sizeof(id));
    const ncclResult_t r = rccl().init_rank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { c->comm = nullptr; c->err = std::string("ncclCommInitRank: ") + rccl().err_string(r); return SCFGP_EHIP; }
    c->comm_ranks = nranks; c->comm_rank = rank;
    if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
    if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
    return SCFGP_OK;
}
extern "C" int scfgp_comm_destroy(scfgp_ctx* c) {
    if (!c) return SCFGP_EARG;
    if (c->comm) {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->st));
        rccl().destroy(c->comm); c->comm = nullptr; c->comm_ranks = 0;
    }
    return SCFGP_OK;
}

extern "C" int scfgp_get_dims(scfgp_ctx* c, int64_t* out, int n) {
    if (!c || !out || n < 6) return SCFGP_EARG;
    out[0] = c->g.K; out[1] = c->g.Kp; out[2] = c->g.Jp; out[3] = c->g.Dp; out[4] = c->g.Np; out[5] = c->g.P;
    if (n >= 7) out[6] = c->g.tile;
    return SCFGP_OK;
}

extern "C" int scfgp_set_profiling(scfgp_ctx* c, int enable) {
    if (!c) return SCFGP_EARG;
    c->prof = enable != 0; c->recs.clear(); c->pool_used = 0;
    return SCFGP_OK;
}

extern "C" int scfgp_get_timings(scfgp_ctx* c, double* ms, const char** names, int n) {
    if (!c) return SCFGP_EARG;
    hipStreamSynchronize(c->st);
    int k = 0;
    for (const ProfRec& r : c->recs) {
        if (k >= n) break;
        float t = 0;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) t = -1;
        if (ms) ms[k] = t;
        if (names) names[k] = r.name.c_str();
        ++k;
    }
    return k;
}

// host-only: geometry as scfgp_create derives it (the row-split self-test below runs without a GPU)
static void derive_geom(Geom& g, int D, int S, int M) {
    g.D = D; g.S = S; g.M = M; g.J = S + M; g.K = 2 * g.J; g.P = 3 + D * S + M * S + S + M;
    g.Dp = (int)round_up(D + 1, 16); g.Jp = (int)round_up(g.J, XT);
    g.Sp = (int)round_up(S + 1, 16); g.lowrank = g.Sp < g.Dp;     // F = l_F r_F^T: project through the S columns when that is narrower
    // Gram tile grid: 128-wide tiles cover K in 64-column blocks; an odd block count ends in a 64-high strip
    g.tile = 128;
    g.gfull = (int)(round_up(g.K, 64) / 128);
    g.gstrip = (int)(round_up(g.K, 64) / 64 % 2);
    g.Kp = (g.gfull + g.gstrip) * g.tile;
}
extern "C" int scfgp_selftest_row_splits(int D, int S, int M, int64_t N, int dtype, int nsplit, int taper) {
    if (D < 1 || S < 1 || M < 1 || N < 1) return SCFGP_EARG;
    Geom g{};
    derive_geom(g, D, S, M);
    g.N = N; g.Np = round_up(N, 256);
    const int jobs = dtype == SCFGP_F32 ? SweepKernels<float>::gram_jobs(g) : SweepKernels<double>::gram_jobs(g);
    const RowSplits rs = gram_row_splits(jobs, g.Np, dtype == SCFGP_F32, nsplit, taper);
    if (rs.nsplit < 1 || rs.nsplit != rs.groups * rs.per_group()) return 1;
    int64_t at = 0;
    for (int s = 0; s < rs.nsplit; ++s) {                       // the splits tile [0, Np) in order, on 256-row blocks
        int64_t r0, r1; rs.range(s, r0, r1);
        if (r0 != at || r1 < r0 || r0 % rs.gran || r1 % rs.gran || (rs.gran != 64 && rs.gran != 256)) return 2;
        at = r1;
    }
    return at == g.Np ? 0 : 3;
}

extern "C" int scfgp_set_option(scfgp_ctx* c, const char* name, int64_t value) {
    if (!c || !name) return SCFGP_EARG;
    const std::string s(name);
    if (s == "gram_nsplit") c->gram_nsplit = (int)value;
    else if (s == "gram_taper") c->gram_taper = (int)value;
    else if (s == "apply_dma") { if (value < -1 || value > 2) { c->err = "apply_dma: -1 auto, 0 off, 1 = 128-wide tiles, 2 = 256-wide (fp32)"; return SCFGP_EARG; }
                                 c->apply_dma = (int)value; }
    else if (s == "gram_chunk") c->gram_chunk = value;
    else if (s == "f16_gram") c->f16gram = (int)value;
    else if (s == "lowrank_bwd") c->lowrank_bwd = (int)value;
    else if (s == "xtz_nsplit") c->xtz_nsplit = (int)value;
    else if (s == "use_graph") c->use_graph = (int)value;
    else if (s == "gram64") { if (value < 0 || value > 3) { c->err = "gram64: 0 never, 1 always level 1, 2 auto, 3 always level 2"; return SCFGP_EARG; }
                              c->gram64 = (int)value; c->esc_level = 0; c->esc_denied = 0; }
    else if (s == "factor_form") { if (value < -1 || value > 1) { c->err = "factor_form: -1 auto, 0 never, 1 always"; return SCFGP_EARG; } c->factor_form = (int)value; }
    else if (s == "cond_threshold") c->esc_thr = (double)value;
    else if (s == "cond_threshold_w") c->escw_thr = (double)value;
    else if (s == "roctx") c->roctx_on = value != 0;
    else if (s == "test_deny_level") { c->test_deny_level = (int)value; return SCFGP_OK; }       // tests: levels >= value are refused as if out of memory
    else if (s == "test_fail_stage") { c->test_fail_stage = (int)value; return SCFGP_OK; }       // tests: the next sweep `value` (1..3) fails
    else { c->err = "unknown option " + s; return SCFGP_EARG; }
    // kernels not run so far, buffers not allocated so far: the next scfgp_train starts with an eager iteration again
    c->warm = false; c->cond_valid = false;
    if (c->gexec) { hipGraphExecDestroy(c->gexec); c->gexec = nullptr; }
    if (c->graph) { hipGraphDestroy(c->graph); c->graph = nullptr; }
    if (c->have_data) return ensure_rows(c, c->g.N);
    return SCFGP_OK;
}

// out[0] condition estimate of A = Phi^T Phi + lam I from the last finished evaluation: max_i L_ii^2 * max_j (A^-1)_jj, a
//        LOWER bound of cond_2(A) (>= the diagonal ratio max L_ii^2 / min L_ii^2)
// out[1] precision level that evaluation ran at (0, 1, 2: see scfgp_ctx; fp64 mode reports 0)
// out[2] 1 if its G and Phi^T y were formed in fp64 from fp64 features (always in fp64 mode), else 0
// out[3] predicted relative error of alpha / Li had the Gram been formed in fp32: SCFGP_ERR_PER_COND * out[0] (an order of
//        magnitude: measured / predicted = 0.2 .. 7)
// out[4] threshold of level 1, out[5] threshold of level 2 (auto policy), out[6] min L_ii^2, out[7] max L_ii^2, out[8] max_j (A^-1)_jj
extern "C" int scfgp_get_condition(scfgp_ctx* c, double* out, int n) {
    if (!c || !out || n < 4) return SCFGP_EARG;
    const double v[9] = {c->cond_est(), (double)c->last_level, c->last_used64 ? 1.0 : 0.0, c->alpha_err_fp32(),
                         c->esc_thr, c->escw_thr, c->cond[0], c->cond[1], c->cond[2]};
    for (int i = 0; i < n && i < 9; ++i) out[i] = v[i];
    return SCFGP_OK;
}

extern "C" int64_t scfgp_debug_read(scfgp_ctx* c, const char* name, void* host, int64_t max_bytes) {
    if (!c || !name || !host) return SCFGP_EARG;
    const Geom& g = c->g;
    const std::string s(name);
    if (s == "chol_trace") { if (hipStreamSynchronize(c->st) != hipSuccess) return SCFGP_EHIP; return chol_trace_read(host, max_bytes); }
    if (s == "apply_trace") { if (hipStreamSynchronize(c->st) != hipSuccess) return SCFGP_EHIP; return apply_trace_read(host, max_bytes); }
    if (s == "trace") { if (hipStreamSynchronize(c->st) != hipSuccess) return SCFGP_EHIP; return trace_read(host, max_bytes); }
    const int64_t K2 = (int64_t)g.Kp * g.Kp;
    const size_t ts = c->tsize();
    const void* src = nullptr; int64_t bytes = 0;
    DevTmp tmp;
    if (s == "Phi") { src = c->d_Phi; bytes = ts * g.Np * g.Kp; }
    else if (s == "V") { src = c->d_V; bytes = ts * g.Np * g.Kp; }
    else if (s == "G") {                                         // exchange buffer 1 unpacked into a buffer of its own: d_x1 is the
        if (dmalloc(c, &tmp.p, 8 * c->n_x1)) return SCFGP_EHIP;  // factorisation's working matrix from scfgp_factor on
        unpack_exchange(c, c->d_xp1, tmp.p); src = tmp.p; bytes = 8 * c->n_x1;
    }
    else if (s == "W") { if (c->stage == 3) unpack_exchange(c, c->d_xp2, c->d_x2); src = c->d_x2; bytes = 8 * c->n_x2; }
    else if (s == "XZ") { src = c->d_x3; bytes = 8 * c->n_x3; }
    else if (s == "Li") { src = c->d_Li; bytes = 8 * K2; }
    else if (s == "B") { src = c->d_B; bytes = 8 * K2; }
    else if (s == "Abar") { src = c->d_Abar; bytes = 8 * K2; }
    else if (s == "p") { src = c->d_p; bytes = 8 * g.Np; }
    else if (s == "q") { src = c->d_q; bytes = 8 * g.Np; }
    else if (s == "vecs") { src = c->d_vecs; bytes = 8 * 5 * g.Kp; }
    else if (s == "Fall") { src = c->d_Fall; bytes = 8 * (int64_t)g.Dp * g.Jp; }
    else if (s == "Xt") { src = c->d_Xt; bytes = 8 * g.Np * g.Dp; }
    else if (s == "scalars") { src = c->d_scalars; bytes = 8 * 32; }
    else { c->err = "debug_read: unknown buffer " + s; return SCFGP_EARG; }
    if (!src) return SCFGP_EARG;
    if (bytes > max_bytes) bytes = max_bytes;
    if (hipStreamSynchronize(c->st) != hipSuccess) return SCFGP_EHIP;
    if (hipMemcpy(host, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return SCFGP_EHIP;
    return bytes;
}
