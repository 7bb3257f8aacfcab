// Host-callable launchers for every device kernel of the scfgp HIP library.
// All launch asynchronously on `st`; none allocates or synchronises.
#pragma once
#include "common.h"

// geometry shared by all sweeps (device buffers are padded so kernels need no edge code)
struct Geom {
    int D, S, M, J, K, P;
    int Dp;      // round_up(D+1,16): X~ = [X | 1 | 0..]   (ones column carries the phase offsets)
    int Jp;      // round_up(J,128)
    int Sp;      // round_up(S+1,16): T~ = [X l_F | 1 | 0..] (N x Sp), the rank-S form of the projection
    int lowrank; // 1: Z = (X~ Lall) Rall = T~ Rall (chosen when Sp < Dp); 0: Z = X~ Fall
    int Kp;      // round_up(K,tile): leading dimension of Phi and of every K x K matrix
    int tile;    // 128: square tile of the Gram products and of the packed exchange layout
    int gfull;   // Gram tile grid: gfull rows of square tiles ...
    int gstrip;  // ... plus, for an odd number of 64-column blocks in K, one 64-high strip below them
    int64_t N;   // valid local rows
    int64_t Np;  // round_up(N,256)
};

// Row splits of the Gram products.  A job is (row split, output tile); the launch runs the jobs split-major and the
// XCD map hands each of the 8 XCD groups a contiguous run of splits, so the workgroups that are resident together
// work on the same rows (their re-reads of Phi are served by the Infinity Cache / L2 instead of HBM) while faster CUs
// simply take more jobs.  A launch ends when the last jobs of every group finish: with equal splits those are full-size
// jobs (5.2 ms of a 38.6 ms launch, 3.7 ms of it with part of the chip idle: profiles/r02_tuning.md), so the LAST
// unit of every group is cut into splits of 1/2, 1/4, 1/8, 1/8 of a unit (`taper`): the tail shrinks eightfold for
// three more slabs per group.
struct RowSplits {
    int nsplit;          // all splits
    int groups;          // 8 (one run of splits per XCD group) or 1
    int units;           // equal-size units per group; taper: the last unit is cut into taper + 1 splits
    int taper;           // 0, or t: the last unit of every group is cut into 1/2, 1/4, .., 1/2^t, 1/2^t
    int gran;            // row granule of the split boundaries: 256, or 64 for problems too small to fill the chip otherwise
    int64_t nrb;         // granules: Np / gran
    __host__ __device__ int per_group() const { return units + taper; }
    // rows [r0, r1) of split s (multiples of gran; may be empty)
    __host__ __device__ void range(int s, int64_t& r0, int64_t& r1) const {
        const int pg = per_group(), grp = s / pg, i = s % pg;
        const int64_t g0 = nrb * grp / groups, g1 = nrb * (grp + 1) / groups, len = g1 - g0;
        const int64_t den = (int64_t)1 << taper;                    // position in 2^-taper of a unit
        const int k = i - (units - 1);                              // 0..taper inside the last unit
        int64_t e0, e1;
        if (taper == 0 || k < 0) { e0 = den * i; e1 = e0 + den; }
        else {
            const int64_t base = den * (units - 1);
            e0 = base + den - (den >> k);                           // 0, 1/2, 3/4, ...
            e1 = k == taper ? base + den : base + den - (den >> (k + 1));
        }
        r0 = (g0 + len * e0 / (den * units)) * gran;
        r1 = (g0 + len * e1 / (den * units)) * gran;
    }
};
RowSplits gram_row_splits(int jobs_per_split, int64_t Np, bool f32, int nsplit_override, int taper);

// operands of the phase projection: Fall (Dp x Jp); Lall (Dp x round_up(Sp,64)) = [l_F | e_D], Rall (Sp x Jp) =
// [[I_S | r_F^T]; phase offsets], Tt (Np x Sp) scratch for T~
struct Projection { const double* Fall; const double* Lall; const double* Rall; double* Tt; };

// ---- fmap.hip: feature map and operand staging ---------------------------------------------------------------------
template <typename T>
struct FmapKernels {
    // Phi = s*[cos Z, sin Z], Z = X~ . Fall  or, F = l_F r_F^T being rank S (SCFGP.py:83), Z = (X~ . Lall) . Rall
    //                                                            (SCFGP.py:98-102 / :139-142)
    static void featuremap(const Geom& g, const double* Xt, const Projection& pr, const Scal* sc, T* Phi, hipStream_t st);
    // fp64 Kp x Kp matrix -> sweep operand (type T, rows/cols >= K zeroed)
    static void convert(const double* src, T* dst, int K, int Kp, hipStream_t st);
    static void convert_transposed(const double* src, T* dst, int K, int Kp, hipStream_t st);      // dst = src^T on the K x K block
};

// ---- gram.hip: TN products (contraction over the rows) ---------------------------------------------------------------
template <typename T>
struct GramKernels {
    // lower tiles of  Phi^T diag(w) Phi  into per-split fp64 slabs (SCFGP.py:104; weighted: backward of :111-113)
    // and, from the diagonal tiles, sidepart[split][Kp] = partials of Phi^T side (side = y: SCFGP.py:108)
    //   qhead: 8 ints of device scratch (the heads of the per-XCD job queues of the persistent launch)
    //   ws2:   2 (Np + 32) floats of device scratch (fp32 mode: the packed row weights / side multipliers of the LDS-DMA tiles;
    //          the 64 floats behind the last row are read, never used); unused in fp64 mode
    static void gram(const Geom& g, const T* Phi, const double* w, const double* side, const RowSplits& rs, int64_t chunk,
                     double* slabs, double* sidepart, int* qhead, float* ws2, hipStream_t st);
    static int gram_jobs(const Geom& g);        // workgroups per row split of gram() (sizes the row split)
    // X~^T Zbar into per-split fp64 slabs, Zbar[n][j] = Phi[n][j] Phibar[n][J+j] - Phi[n][J+j] Phibar[n][j]
    // formed inside the operand loader
    static void xtz(const Geom& g, const double* Xt, const T* Phi, const T* Phibar, int nsplit, int64_t chunk, double* slabs,
                    hipStream_t st);
    // ---- rank-S form of the backward projection (F = l_F r_F^T, SCFGP.py:83,100: the reverse sweep needs only
    //      T~^T Zbar (S+1 x J) and X~^T (Zbar_L + Zbar_M r_F) (D+1 x S) instead of X~^T Zbar (D+1 x J)) -------------
    // Zbar over the cosine half of Phibar, in place
    static void zbar_inplace(const Geom& g, const T* Phi, T* Phibar, hipStream_t st);
    // Rsel = [I_S ; r_F] as a typed Kp x Kp-strided operand (round_up(S,64) columns written)
    static void rsel(const Geom& g, const double* params, T* out, hipStream_t st);
    // A^T Bm over the rows into per-split slabs: A (Np x Dp fp64), Bm (Np x J live columns, leading dimension ldb)
    static void tn_plain(const double* A, int Dp, const T* Bm, int64_t ldb, int J, int64_t Np, int nsplit, int64_t chunk, double* slabs,
                         hipStream_t st);
};

// operands of one apply product in compute mode SCFGP_F16X3 (apply_f16.hip): Phi and the K x K operand in plane form (below), scale[0] =
// 2^-(e_Phi + e_operand) on the device.  V16 (may be NULL): V = Phi B's epilogue also writes V in plane form, scaled by 2^e with
// e = 14 - ilogb(vbound[0]) -- the array split_v would otherwise write
struct F16Operands { const unsigned* Phi16; const char* B16; const float* scale; unsigned* V16; const float* vbound; };

// ---- apply.hip: NT products (contraction over the feature columns) and the per-row statistics ------------------------
//   dma: 0 = operands staged through registers; 1 / 2 = the full 128-column tiles by LDS-DMA, 128 / 256 wide (256: fp32 only)
template <typename T>
struct ApplyKernels {
    // V = Phi . Bm, vpart[jt][n] = sum_{j in tile} Phi[n][j] V[n][j]    (SCFGP.py:112); column tile jt also forms
    // mupart[jt][n] = its slice of mu = Phi . alpha (SCFGP.py:111 / :143) from the rows it stages
    //   f16 (fp32 mode only, dma == 2): the 256-wide tiles by the three-term fp16 split instead (apply_f16.hip); the narrower tiles of
    //   the launch plan stay exact fp32
    static void apply_v(const Geom& g, const T* Phi, const T* Bm, T* V, double* vpart, const double* alpha, double* mupart,
                        hipStream_t st, int dma = 0, const F16Operands* f16 = nullptr);
    // predict: vpart[jt][n] = slices of v_n = || Li phi_n ||^2 (the reference's rowsum((Phi Li^T)^2), SCFGP.py:144) from the
    // triangular product Phi . LiT, LiT[k][j] = Li[j][k] (convert_transposed); mupart as apply_v.  Half the flops of apply_v.
    static void apply_predict(const Geom& g, const T* Phi, const T* LiT, double* vpart, const double* alpha, double* mupart,
                              hipStream_t st);
    // Factor form of pass 2 (the reference's own products, SCFGP/SCFGP.py:112: v = rowsum((Phi Li^T)^2)): C = Phi . LiT
    // (triangular: half the flops), vpart = slices of rowsum(C^2), mupart = slices of mu = Phi . alpha = C . beta (SCFGP.py:109-111:
    // alpha = Li^T beta, beta = Li Phi^T y; the LDS-DMA tiles use the second form); then V = C . Li = Phi B (triangular).
    // Li / LiT: the typed K x K copies of L^-1 and its transpose (padding zeroed).  Rounding errors of C are amplified by
    // cond(L) = sqrt(cond(A)) where those of V = Phi . B computed directly are amplified by cond(A)
    // (profiles/r03_c3_owner.md).
    static void apply_c(const Geom& g, const T* Phi, const T* LiT, const T* Li, T* C, double* vpart, const double* alpha,
                        const double* beta, double* mupart, hipStream_t st, int dma = 0);
    static void apply_vc(const Geom& g, const T* C, const T* Li, const T* LiT, T* V, hipStream_t st, int dma = 0);
    // Phibar = 2 Phi.Abar + 2 q V + p alpha^T + y ut^T  (in place over V)
    static void apply_phibar(const Geom& g, const T* Phi, const T* Abar, T* V, const double* p, const double* q,
                             const double* y, const double* alpha, const double* ut, hipStream_t st, int dma = 0, const F16Operands* f16 = nullptr);
    // Out = A . Bm over k < Kc for ncols columns (64-wide tiles, leading dimension Kp everywhere; Bm[k][c] = 0 for k < c)
    static void apply_plain(const Geom& g, const T* A, const T* Bm, T* Out, int Kc, int ncols, hipStream_t st);
    // per-row moments and adjoint scalars; block partials (4 per block) of T2, kbar, sum q v, sum p mu  (SCFGP.py:111-113,121-124)
    static void rowstats(const Geom& g, const double* mupart, const double* vpart, const double* y,
                         const Scal* sc, double* p, double* q, double* partial, int nblocks, hipStream_t st);
    // predictive mean / std                                          (SCFGP.py:143-144)
    static void rowpredict(const Geom& g, const double* mupart, const double* vpart, const Scal* sc, double* mu, double* sd,
                           hipStream_t st);
};

// ---- apply_f16.hip, gram_f16.hip: compute mode SCFGP_F16X3 -- the big products as a three-term fp16 split (a labelled secondary mode) ----
// "Plane form": 4 bytes per element, per 16 consecutive columns the 16 h's and then the 16 l's ([16 x h | 16 x l], 64 bytes).  Operands of
// one apply product: Phi in plane form (Np x Kp), the K x K operand in plane form (row = output column), scale[0] = 2^-(e_Phi + e_operand)
// on the device
struct F16x3Kernels {
    // M: fp64, symmetric, K x K inside Kp x Kp; part: >= 512 doubles of scratch
    static void split_operand(const Geom& g, const double* M, char* B16, float* scale, double* part, const Scal* sc, hipStream_t st);
    // column tiles [col0, col0 + BN njt) (BN = 256, 128 or 64; vpart slots from slot0) of row blocks rb0 .. rb0 + nrb - 1, epilogue EPI 0
    // (V, row dots) or 1 (Phibar); returns the tile count
    template <int EPI, int BN>
    static int apply(const Geom& g, int njt, int col0, int slot0, const float* Phi, const F16Operands& f, float* V,
                     double* vpart, const double* p, const double* q, const double* y, const double* alpha, const double* ut, double* mu,
                     hipStream_t st, int64_t rb0, int64_t nrb);
    // gram_f16.hip.  The Np x Kp plane-form arrays are allocated with F16_PAD bytes behind them (the Gram's last 256-column block may
    // stick out of Kp).  tmp: 8 floats on the device; after a split pass tmp + 4 is the scale of the Gram product that follows.
    // sidepart: side_blocks(g) x Kp doubles, the block partials of M^T w (reduce_side sums them).
    static constexpr size_t F16_PAD = 1024;
    static int side_blocks(const Geom& g);
    // Phi -> plane form (Phi16: the operand of the apply tiles and of pass 1's Gram); sidepart <- Phi^T y
    static void split_phi(const Geom& g, const float* Phi, const double* y, const Scal* sc, unsigned* Phi16, double* sidepart, float* tmp,
                          hipStream_t st);
    // tmp[5] <- a bound of |V| = |Phi B| from B alone (fp64, symmetric, ld Kp): before the product, whose epilogue writes V's planes with it
    static void v_bound(const Geom& g, const double* B, const Scal* sc, float* tmp, hipStream_t st);
    // V = Phi B -> plane form of diag(q) V (and of V itself unless V16g is NULL: the product's epilogue wrote it); sidepart <- V^T p
    static void split_v(const Geom& g, const float* V, const double* q, const double* p, unsigned* V16g, unsigned* qV16g, double* sidepart,
                        float* tmp, hipStream_t st);
    // slabs[chunk][tri(128-tile)][128 x 128] = scale[0] (A chunk)^T (B chunk) on the lower 128-tiles, every slab written; chunk rows per
    // chunk (rounded up to 256): gram_chunks(g, chunk) chunks, to be summed by reduce_tri_tiles with nsplit = that number
    static int gram_chunks(const Geom& g, int64_t chunk);
    static int gram_tiles(const Geom& g);                       // 256 x 128 tiles of one chunk
    static void gram(const Geom& g, const unsigned* A16g, const unsigned* B16g, const float* scale, int64_t chunk, double* slabs, hipStream_t st);
};

// everything that sweeps the rows, under one name
template <typename T> struct SweepKernels : FmapKernels<T>, GramKernels<T>, ApplyKernels<T> {};

// diagnostic builds only (-DSCFGP_TRACE): per-workgroup [start, end, xcc, kind] of the last Gram launch; -1 otherwise
int64_t trace_read(void* host, int64_t max_bytes);
int64_t apply_trace_read(void* host, int64_t max_bytes);    // per workgroup of the last LDS-DMA apply launch: [start, end, xcc]
int64_t chol_trace_read(void* host, int64_t max_bytes);     // per Cholesky step: 12 phase stamps of workgroup 0

// ---- reductions ------------------------------------------------------------
// packed lower tiles = sum over splits of the per-split lower-tile slabs (tile t = ti(ti+1)/2+tj, row-major)
void reduce_tri_tiles(const double* slabs, int nsplit, int nts, int tile, double* packed, hipStream_t st);
// packed lower tiles -> full symmetric Kp x Kp matrix
void unpack_tri_tiles(const double* packed, int nts, int tile, double* full, int64_t ld, hipStream_t st);
// out (ldo) = sum over splits of a full ntm x ntn tile grid of slabs
void reduce_full_tiles(const double* slabs, int nsplit, int ntm, int ntn, double* out, int64_t ldo, hipStream_t st);
// vec[j < ncov] = sum over splits of sidepart[split][j] (ld Kp), vec[ncov..Kp) = 0
void reduce_side(const double* sidepart, int nsplit, int Kp, int ncov, double* vec, hipStream_t st);
// scalars[slot0 + k] = sum_b partial[b*width + k], k < width (the partials are scratch: a large single-scalar sum is staged in place)
void reduce_scalars(double* partial, int nblocks, int width, double* scalars, int slot0, hipStream_t st);
// scalars[slot] = sum y^2
void sum_squares(const double* y, int64_t n, double* scalars, int slot, double* scratch, hipStream_t st);

// ---- parameter unpack / gradient epilogue -----------------------------------
void unpack_params(const Geom& g, const double* params, double* F, double* Fall, double* Lall, double* Rall, Scal* sc, hipStream_t st);
//   rank-S form (TZ != NULL): XZ is not used; TZ (ld ldtz) rows s < S hold l_F^T X^T Zbar and row S the column sums of Zbar,
//   XU (ld ldxu) holds X~^T (Zbar_L + Zbar_M r_F)
void grad_epilogue(const Geom& g, const double* params, const double* F, const double* XZ, int64_t ldxz,
                   double* work, double* scalars, int64_t Nglobal, double* grad, hipStream_t st,
                   const double* TZ = nullptr, int64_t ldtz = 0, const double* XU = nullptr, int64_t ldxu = 0);
// yy -> y^T y, t2kb -> (T2, kbar, sum q v, sum p mu): device scalars living in the exchange buffers (summed over ranks)
void finalize_cost(const Geom& g, const Scal* sc, double* scalars, const double* yy, const double* t2kb,
                   int64_t Nglobal, double* grad, int want_grad, hipStream_t st);
// xs[XS_RAN1 .. XS_FAIL] = the four status slots of an exchange buffer's scalar tail (common.h)
void write_status(double* xs, double ran1, double cap1, double cap2, double fail, hipStream_t st);
// Xt (Np x Dp) = [X[idx] | 1 | 0], zero rows >= N; y padded with zeros; idx == NULL: rows in order
// mode/sp: optional per-column input scaling (SCFGP/Scaler.py forward_transform) applied on the fly
void pack_data(const Geom& g, const double* Xraw, const double* yraw, const int64_t* idx, double* Xt, double* y, hipStream_t st,
               int mode = 0, const double* sp = nullptr);
// square K x K host-layout matrix (ld K) -> Kp x Kp with identity padding, and back
void pad_square(const double* src, int K, int Kp, double* dst, hipStream_t st);

// ---- prediction post-processing (SCFGP/SCFGP.py:281-293, SCFGP/Scaler.py:118-135) -------------
constexpr int YPOST_BLOCKS = 64;
// out[0] = mean(ys)
void ypost_mean(const double* ys, int64_t n, double* out, hipStream_t st);
// mu, sd (n) <- y-scaler backward transform of the mean and half the transformed +-1 std band, in place;
// ys != NULL: part[YPOST_BLOCKS][4] = block partials of the metric sums of this chunk
void ypost_chunk(double* mu, double* sd, const double* ys, int64_t n, int mode, const double* sp, const double* ymean,
                 double* part, hipStream_t st);
// out[6] = MAE, NMAE, MSE, NMSE, MNLP, SCORE from nparts x 4 partial sums over n targets
void ypost_metrics(const double* part, int nparts, int64_t n, double* out, hipStream_t st);

// ---- on-device update rules (SCFGP/Optimizer.py) --------------------------------------------
struct OptHyper { double lr, b1, b2, eps, momentum; };   // b1 doubles as rho for rmsprop/adadelta; momentum < 0: no Nesterov
// theta <- rule(theta, grad); st = [s1 | s2 | velocity]; tctr[0] = step counter, tctr[1] = index into hist
void opt_update(int algo, const OptHyper& h, int P, double* theta, const double* grad, double* st, double* tctr,
                const double* scalars, double* hist, int hist_cap, hipStream_t stream);

// ---- K x K stage (fp64, all matrices Kp x Kp, leading dimension Kp) -----------
struct KStage {
    int K, Kp;
    double *A;      // in: G + lam I (full, symmetric); working matrix of the factorisation (the factor L goes to T2)
    double *Li;     // L^{-1} (lower, zero above)
    double *B;      // Li^T Li
    double *T1, *T2;// scratch Kp x Kp
    double *g, *beta, *alpha, *h, *u, *ut;   // Kp vectors
    double *scalars; int* flag;
};
// SCFGP.py:105-110,125; packed: the summed packed lower 128 x 128 tiles of G (exchange buffer 1); the upper blocks of Li stay as
// they are (zero from context creation: nothing ever writes them)
void kstage_factor(const KStage& k, const double* packed, const Scal* sc, hipStream_t st);
void kstage_adjoint(const KStage& k, const double* BWB, double* Abar, const Scal* sc, hipStream_t st);
void kstage_adjoint_factor_form(const KStage& k, double* McBWB, double* Abar, const Scal* sc, hipStream_t st);
// scalars[R_TRAG] = tr(Abar G), scalars[R_UTG] = ut^T Phi^T y from the summed packed G of exchange buffer 1 (after the adjoint)
void kstage_bbar(const KStage& k, const double* packed, const double* Abar, double* part, hipStream_t st);
