/*
 * scfgp_hip.h -- C ABI of libscfgp_hip.so: the MI355X (gfx950) implementation of the
 * SCFGP Fourier-feature marginal-likelihood hot path.
 *
 * The library stands behind the reference's "compiled function triple"
 *     train_func(X,y)       -> [cost, alpha, Li]              SCFGP/SCFGP.py:132-135
 *     train_iter_func(X,y)  -> same outputs + parameter update SCFGP/SCFGP.py:136-137
 *     pred_func(Xs,alpha,Li)-> [mu, std]                       SCFGP/SCFGP.py:138-148
 * i.e. it replaces what theano.function compiled from build_theano_models
 * (SCFGP/SCFGP.py:92-148, gradient from TT.grad at :129).  The optimiser update rule
 * (SCFGP/Optimizer.py) stays on the host above this boundary so its callback signature
 * remains user-extensible; the library returns the exact gradient it needs.
 *
 * Conventions
 *   - every function returns 0 or a negative error code (scfgp_finish and scfgp_factor may also return SCFGP_REDO = 1);
 *     nothing throws across the ABI;
 *     scfgp_last_error() gives a message for the last failure on that context
 *   - the caller owns every host buffer; the library owns all device memory
 *   - host matrices are row-major, C-contiguous float64 regardless of compute dtype
 *     (the reference's graph is float64 throughout: TT.dmatrices, SCFGP/SCFGP.py:95-96,138)
 *   - a context is bound to one GPU and is not thread-safe; calls are synchronous unless
 *     stated otherwise (results are on the host when the call returns)
 *   - K = 2*(S+M) is the Gram dimension, P = 3 + D*S + M*S + S + M the parameter count
 *     (SCFGP/SCFGP.py:72)
 */
#ifndef SCFGP_HIP_H
#define SCFGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct scfgp_ctx scfgp_ctx;

#define SCFGP_OK            0
#define SCFGP_EARG         -1   /* bad argument / wrong call order                       */
#define SCFGP_EHIP         -2   /* HIP runtime error                                     */
#define SCFGP_ENOTPD       -3   /* A = Phi^T Phi + (e^{2a}+1e-6) I not positive definite:
                                   the reference raises numpy.linalg.LinAlgError from its
                                   Cholesky op (SCFGP/SCFGP.py:106)                      */
#define SCFGP_ENONFINITE   -4   /* cost is NaN/Inf                                       */
#define SCFGP_EPEER        -5   /* row shards: ANOTHER rank failed in this evaluation (its exchange
                                   buffers carried the failure mark, scfgp_fail_stage): every rank
                                   returns this from scfgp_finish / scfgp_eval / scfgp_train; the
                                   results are not valid, the context stays usable            */
#define SCFGP_REDO          1   /* not an error, staged API only.  scfgp_finish: the evaluation
                                   ran at a lower precision level than its own condition estimate
                                   asks for; the level has been raised.  scfgp_factor: the ranks
                                   have just agreed on a lower level than some of them ran pass 1
                                   at.  Either way: run the stages again from scfgp_pass1
                                   (scfgp_eval / scfgp_eval_rows do that themselves)      */

#define SCFGP_F64 0             /* fp64 MFMA everywhere (reference numerics)             */
#define SCFGP_F32 1             /* N-sized products in exact-fp32 MFMA, fp64 projection,
                                   fp64 cross-chunk accumulation and fp64 K x K stage    */
#define SCFGP_F16X3 2           /* SECONDARY mode, never what a benchmark headline quotes: SCFGP_F32 in everything but the four
                                   N-sized products (V = Phi B, Phibar = 2 Phi Abar + ..., Phi^T Phi, V^T diag(q) V), which run as a
                                   THREE-TERM fp16 split on the fp16 matrix pipe wherever fp32 mode uses its 256-wide LDS-DMA tiles
                                   (K >= 1024, >= 65536 rows) at precision level 0: x = (h + l) 2^-e, h, l fp16, one power-of-two
                                   scale per operand matrix; h.h + l.h + h.l in fp32 accumulators (fp16 products are exact in
                                   fp32; the Gram folds its accumulators every 512 rows because the instruction truncates).
                                   Errors within 4x of SCFGP_F32's (scfgp_amd/csrc/apply_f16.hip, gram_f16.hip); it runs fp32
                                   mode's parity tier.  Option "f16_gram" = 0 keeps the fp32 Gram in this mode               */

/* ---- life cycle --------------------------------------------------------------------
 * Replaces SCFGP.build_theano_models (SCFGP/SCFGP.py:92-148): "compile" becomes "create a
 * context".  D,S,M as in SCFGP.__init__/set_data (SCFGP/SCFGP.py:36-37,164).
 * `stream` is a hipStream_t (NULL = the context creates its own); passing the stream of
 * the host framework keeps the library's work ordered with that framework's collectives.
 * On failure (negative return) *out is still set whenever the context object itself could be made -- so that
 * scfgp_last_error(*out) can say which allocation or HIP call failed -- and holds whatever was allocated up to that point:
 * the caller must hand it to scfgp_destroy (scfgp_amd/engine.py does). */
int  scfgp_create(scfgp_ctx** out, int D, int S, int M, int dtype, int device, void* stream);
void scfgp_destroy(scfgp_ctx* ctx);
const char* scfgp_last_error(const scfgp_ctx* ctx);

/* ---- implicit state: the shared parameter vector -------------------------------------
 * The reference binds hyper-parameters with givens=[(params, self.params)]
 * (SCFGP/SCFGP.py:134-137,147-148): they are state of the compiled functions, not
 * arguments.  Layout [a,b,c,l_f(D*S),r_f(M*S),l_p(S),p(M)] (SCFGP/SCFGP.py:72,74-90). */
int scfgp_set_params(scfgp_ctx* ctx, const double* params, int P);
int scfgp_get_params(scfgp_ctx* ctx, double* params, int P);

/* ---- training data: upload once, keep resident ----------------------------------------
 * X (N,D), y (N) already scaled (what SCFGP.set_data stores in self.X/self.y,
 * SCFGP/SCFGP.py:161-162).  n_global is the N the objective divides by and uses in
 * 2(N-M)a (X.shape[0] at SCFGP/SCFGP.py:126,128): equal to N on one GPU, the sum over
 * ranks when rows are sharded. */
int scfgp_set_data(scfgp_ctx* ctx, const double* X, const double* y, int64_t N, int64_t n_global);

/* ---- train_func / train_iter_func ------------------------------------------------------
 * One evaluation at the current parameters.  X==NULL uses the resident rows; otherwise
 * (minibatches, SCFGP/SCFGP.py:226-235) the given rows replace them first.
 * want_grad=0 is train_func (forward only); want_grad=1 additionally returns
 * d cost / d params (P) -- what TT.grad (SCFGP/SCFGP.py:129) feeds the update rule.
 * Outputs (any may be NULL): cost (1), grad (P), alpha (K), Li (K*K row-major, lower
 * triangular, zeros above the diagonal). */
int scfgp_eval(scfgp_ctx* ctx, const double* X, const double* y, int64_t N, int want_grad,
               double* cost, double* grad, double* alpha, double* Li);

/* ---- minibatches by index list ----------------------------------------
 * The reference's minibatch loop (SCFGP/SCFGP.py:172-182,226-235) fancy-indexes host copies of X, y
 * and re-passes them; here the batch is a gather of n rows of the RESIDENT data set on the device.
 * Same per-batch semantics (N := n in 2(N-M)a and /N).  A later scfgp_eval(X=NULL) sees all rows again. */
int scfgp_eval_rows(scfgp_ctx* ctx, const int64_t* idx, int64_t n, int want_grad,
                    double* cost, double* grad, double* alpha, double* Li);

/* ---- pred_func  (SCFGP/SCFGP.py:138-148) ------------------------------------------------
 * mu (T), std (T) for Xs (T,D) given alpha (K) and Li (K*K) as returned by scfgp_eval. */
int scfgp_predict(scfgp_ctx* ctx, const double* Xs, int64_t T, const double* alpha, const double* Li,
                  double* mu, double* std);

/* ---- predict on unscaled inputs (input side) -------------------------------
 * SCFGP.predict applies X_scaler.forward_transform on the host before pred_func
 * (SCFGP/SCFGP.py:279, SCFGP/Scaler.py:99-116): min-max, Box-Cox, z-score / normal CDF per column.
 * With the fitted per-column parameters registered once (D doubles each; NULL = unused by `mode`:
 * 0 none, 1 min-max, 2 normal, 3 inv-normal, 4 auto-normal, 5 auto-inv-normal) scfgp_predict_raw
 * takes the column-selected raw Xs and applies the transform inside its packing kernel. */
int scfgp_set_x_scaler(scfgp_ctx* ctx, int mode, const double* min, const double* max, const double* boxcox,
                       const double* mu, const double* std);
int scfgp_predict_raw(scfgp_ctx* ctx, const double* Xs_raw, int64_t T, const double* alpha, const double* Li,
                      double* mu, double* std);
/* The rest of SCFGP.predict on the device (SCFGP/SCFGP.py:281-293, SCFGP/Scaler.py:118-135): with the
 * fitted single-column y scaler registered (same mode numbers), scfgp_predict_y returns
 * mu_y = backward(mu_f) and std_y = (backward(mu_f + std_f) - backward(mu_f - std_f)) / 2 (T doubles each)
 * and, when raw targets ys (T) are given, metrics[6] = MAE, NMAE, MSE, NMSE, MNLP, SCORE.  The X scaler
 * registered with scfgp_set_x_scaler (or none) is applied to Xs_raw as in scfgp_predict_raw. */
int scfgp_set_y_scaler(scfgp_ctx* ctx, int mode, double min, double max, double boxcox, double mu, double std);
int scfgp_predict_y(scfgp_ctx* ctx, const double* Xs_raw, int64_t T, const double* alpha, const double* Li,
                    const double* ys, double* mu_y, double* std_y, double* metrics);

/* ---- staged evaluation for row-sharded data parallelism ---------------------------------
 * The objective needs three row sweeps separated by two K x K stages; with rows sharded
 * over ranks each sweep ends in one sum over ranks.  The host framework (torch.distributed
 * over RCCL) performs that sum in place on the device buffer scfgp_exchange() exposes:
 *
 *   scfgp_pass1   -> exchange 1 = [G, packed lower 128x128 tiles | Phi^T y (Kp) | y^T y ...]
 *   scfgp_factor     (replicated: Cholesky, Li, alpha, log det)
 *   scfgp_pass2   -> exchange 2 = [B W B = V^T diag(q) V, packed lower tiles | B Phi^T p = V^T p (Kp) | T2, kbar, sum q v, sum p mu ...]
 *                    (at precision level 2 of fp32 mode, scfgp_get_condition: [C^T diag(q) C | C^T p | ...], C = Phi Li^T --
 *                    still row sums, so the sum over ranks is the same operation)
 *   scfgp_adjoint    (replicated: Abar)                      [want_grad only]
 *   scfgp_pass3   -> exchange 3 = [X~^T Zbar | ...]          [want_grad only]  (d cost / d b is formed in closed form from the
 *                    summed exchanges 1 and 2: 2 tr(Abar G) + ut^T Phi^T y + 2 sum q v + sum p mu -- no row sweep, no slot here)
 *   scfgp_finish  -> outputs on the host
 *
 * These calls only enqueue work on the context's stream (asynchronous); scfgp_finish
 * synchronises.  scfgp_eval is exactly this sequence without the sums.
 *
 * Ranks decide together.  The last four of the 8 scalars that close every exchange buffer are a status word: each rank
 * writes 0 / 1 there, the sum over ranks turns them into COUNTS that every rank reads alike --
 *   [4] ranks whose pass 1 ran at precision level >= 1        [5] / [6] ranks that cannot reach level 1 / 2 (their row
 *   buffers were refused)            (exchange 1)             [7] ranks that failed in this sweep (exchanges 1, 2, 3).
 * After a precision level was raised (the decision comes from the summed matrix, so every rank tries in the same
 * evaluation) scfgp_factor reads the summed word once -- one stream synchronisation in that evaluation only -- and every
 * rank commits to the lowest level any rank can reach; if that changes the form of pass 1 on some rank, scfgp_factor
 * returns SCFGP_REDO on ALL ranks.  scfgp_get_condition()[1] then reports the common level, scfgp_last_error the refusal.
 * A rank whose sweep fails calls scfgp_fail_stage for that and every later exchange of the evaluation (stage 1..3; stage 3
 * only with want_grad): the buffer is marked, the sum still happens (inside the library with a communicator, else by the
 * host framework as usual), nobody waits in a collective for a rank that has left, and scfgp_finish returns SCFGP_EPEER on
 * every other rank.  scfgp_eval / scfgp_eval_rows / scfgp_train do this themselves when the sums run inside the library. */
int scfgp_pass1(scfgp_ctx* ctx);
int scfgp_factor(scfgp_ctx* ctx);
int scfgp_pass2(scfgp_ctx* ctx, int want_grad);
int scfgp_adjoint(scfgp_ctx* ctx);
int scfgp_pass3(scfgp_ctx* ctx);
int scfgp_finish(scfgp_ctx* ctx, int want_grad, double* cost, double* grad, double* alpha, double* Li);
int scfgp_fail_stage(scfgp_ctx* ctx, int stage, int want_grad);
/* optional, any time after scfgp_factor, best after the remaining stages are queued: copies alpha (K) and
 * Li (K*K) to the host on a second stream through pinned staging and returns once they are in the caller's
 * arrays.  It waits for the factor stage only, so the transfer and the host-side copy overlap passes 2 and 3
 * (then pass NULL for both to scfgp_finish). */
int scfgp_fetch_factors(scfgp_ctx* ctx, double* alpha, double* Li);
/* device pointer + length (in doubles) of exchange buffer `stage` (1..3) */
int scfgp_exchange(scfgp_ctx* ctx, int stage, void** dev_ptr, int64_t* count);
/* Ordering contract for the sums.  The library enqueues on ITS stream: the one handed to scfgp_create, or a
 * private one when that was NULL (torch's default stream is handle 0 == NULL, so a caller on torch's default
 * stream always gets a private library stream).  A collective issued by the host framework on another stream
 * must be fenced on both sides:
 *     scfgp_stream_fence(ctx, peer, 0)   peer waits for the library's queued work   (before the all-reduce)
 *     scfgp_stream_fence(ctx, peer, 1)   the library waits for peer's queued work   (after the all-reduce)
 * peer = the hipStream_t the collective runs on (NULL = legacy default stream).  Event based, asynchronous,
 * a no-op when peer is the library's own stream.  No reference counterpart (the reference is single-device). */
int scfgp_stream_fence(scfgp_ctx* ctx, void* peer_stream, int direction);

/* ---- the sums inside the library: RCCL all-reduce over xGMI (no reference counterpart: the reference is single-device) ----
 * One process per GPU.  Rank 0 calls scfgp_comm_unique_id and hands the 128 bytes to the other ranks by any means (MPI, a file,
 * torch.distributed.broadcast_object_list); every rank then calls scfgp_comm_init on its context (collective: ncclCommInitRank).
 * From then on the three sums of the row sums of SCFGP/SCFGP.py:104,108,126 and of the reverse sweep -- exchange buffers 1..3
 * above -- are ncclAllReduce(fp64, sum) calls the library enqueues itself on the context's stream at the end of scfgp_pass1 /
 * scfgp_pass2 / scfgp_pass3, so scfgp_eval and scfgp_eval_rows are complete sharded evaluations (set the rows of the rank with
 * scfgp_set_data(..., n_global = sum of the ranks' N)); the precision level is decided from the summed matrix and committed
 * through the status word above, alike on every rank.  scfgp_train runs its iterations with the three sums inside (every rank
 * applies the same deterministic rule to the same summed gradient: the parameter vectors stay bit-equal without a broadcast;
 * eager launches by default, the captured graph only with option "use_graph" = 2).  librccl.so is looked up at run time (a copy
 * the process already carries is reused, wherever it was loaded from): the library has no link-time dependency on it and
 * single-GPU users never load it.  A caller that prefers to run the sums in its own framework leaves the communicator out and
 * uses scfgp_exchange + scfgp_stream_fence as before.
 * STATUS: the communicator path has run on hardware with ONE rank only (the build pool hands out single GPUs); with two or
 * more ranks it is covered by construction and by the in-process / gloo rehearsals of the same protocol, not by a measurement. */
int scfgp_comm_unique_id(void* id128);
int scfgp_comm_init(scfgp_ctx* ctx, int nranks, int rank, const void* id128);
int scfgp_comm_destroy(scfgp_ctx* ctx);

/* ---- on-device update rule and multi-iteration residency -------------
 * The arithmetic of SCFGP/Optimizer.py as a device kernel behind the evaluation, so a training
 * iteration (SCFGP/SCFGP.py:237: train_iter_func) needs no host round trip; from the second
 * iteration on the whole iteration is ONE captured hipGraph launch.
 *   algo   0 sgd, 1 adagrad, 2 rmsprop, 3 adadelta, 4 adam, 5 adamax  (SCFGP/Optimizer.py:99-382)
 *   hyper  [learning_rate, beta1 (rho for rmsprop/adadelta), beta2, epsilon]
 *   momentum  Nesterov momentum as the reference applies it (SCFGP/Optimizer.py:62-97, on the FIRST
 *             state of the rule's update dictionary); negative = none
 * scfgp_train runs n_iters x (evaluate + update) on the resident rows (of this rank: see the communicator above), returns the cost
 * of every iteration (each at its pre-update parameters, like train_iter_func) and, if non-NULL,
 * alpha / Li of the LAST evaluation; the updated vector is read with scfgp_get_params.
 * scfgp_opt_state copies optimiser state to (set=0) or from (set=1) the host: which 0,1 = the rule's
 * two state vectors, 2 = Nesterov velocity (P doubles each), 3 = step counter (1 double). */
int scfgp_opt_init(scfgp_ctx* ctx, int algo, const double* hyper, int nhyper, double momentum);
int scfgp_opt_state(scfgp_ctx* ctx, int set, int which, double* buf);
/* one step of the device rule with a caller-supplied gradient (P doubles, host): parameters and state advance as inside
 * scfgp_train, no evaluation (for gradients formed elsewhere; the rule's known-answer tests drive it) */
int scfgp_opt_step(scfgp_ctx* ctx, const double* grad, int P);
int scfgp_train(scfgp_ctx* ctx, int n_iters, double* cost_hist, double* alpha, double* Li);

/* ---- conditioning and precision level (no reference counterpart) -----------------------------------
 * The reference computes in float64 throughout (SCFGP/SCFGP.py:95-96) and factors A = Phi^T Phi + (e^{2a}+1e-6) I
 * (SCFGP/SCFGP.py:104-107) whatever its conditioning.  In SCFGP_F32 mode the Gram products carry a
 * relative error of ~6e-8, which reaches alpha and Li multiplied by the condition of A (measured: about 3e-7 times the
 * estimate below).  The K x K stage therefore reports a condition estimate with every evaluation, and by default (option
 * "gram64" = 2, auto) the library raises the precision of the two Gram products when it is high:
 *   level 1 (estimate > 10):   pass 1 -- G and Phi^T y by the fp64 kernels from fp64 features; alpha and Li then equal fp64
 *                               mode's bit for bit, cost / mu* / sigma* to ~1e-8
 *   level 2 (estimate > 1000): also pass 2 in the reference's own factor form (SCFGP/SCFGP.py:112): C = Phi Li^T,
 *                               v = rowsum(C^2), V = C Li, all in fp32 MFMA; EXCHANGE BUFFER 2 THEN CARRIES C^T diag(q) C AND
 *                               C^T p (not B W B and u): the K x K stage forms B W B = Li^T (C^T diag(q) C) Li and
 *                               u = Li^T (C^T p) after the sum over ranks.  Rounding errors are amplified by sqrt(cond A)
 *                               instead of cond A (gradient blocks to ~1e-4)
 * "gram64" = 0 never (plain fp32: the caller reads out[3] to know what alpha is worth), 1 / 3 = always level 1 / 2.
 * out (n >= 4, up to 9 values): [0] condition estimate max_i L_ii^2 * max_j (A^-1)_jj of the last finished evaluation (a lower
 * bound of cond_2(A)), [1] level it ran at, [2] 1 if G was formed in fp64 from fp64 features, [3] predicted relative error
 * of alpha / Li for an fp32 Gram at this estimate, [4] / [5] thresholds of levels 1 / 2, [6..8] min L_ii^2, max L_ii^2,
 * max_j (A^-1)_jj. */
int scfgp_get_condition(scfgp_ctx* ctx, double* out, int n);

/* ---- introspection ------------------------------------------------------------------------ */
/* padded sizes the device buffers use: out[0]=K, out[1]=Kp, out[2]=Jp, out[3]=Dp, out[4]=Np, out[5]=P, out[6]=tile */
int scfgp_get_dims(scfgp_ctx* ctx, int64_t* out, int n);
/* profiling: enable per-stage hipEvent timing; after an evaluation read back up to n
 * (name, milliseconds) pairs.  Returns the number of stages recorded. */
int scfgp_set_profiling(scfgp_ctx* ctx, int enable);
int scfgp_get_timings(scfgp_ctx* ctx, double* ms, const char** names, int n);
/* copy an internal device buffer to the host for tests ("Phi","V","G","W","XZ","Li","B","Abar",
 * "p","q","vecs","Fall","Xt","scalars"); returns the number of bytes copied or <0.  "G" is exchange buffer 1 unpacked
 * (the summed Gram, Phi^T y, y^T y and the status word) at any stage; "W" is exchange buffer 2 as the adjoint stage left it */
int64_t scfgp_debug_read(scfgp_ctx* ctx, const char* name, void* host, int64_t max_bytes);
/* options (name, value):
 *   "gram_nsplit"  row-split units of the Gram products (0 = default)
 *   "gram_taper"   1: the last unit of every XCD group is cut into 1/2, 1/4, 1/8, 1/8; t >= 2: into t + 3 pieces down to 1/2^(t+2)
 *   "gram_chunk"   fp32 mode: rows between two flushes of the fp32 accumulators into the fp64 slabs (default 4096; f16x3 mode:
 *                  half the rows of a Gram job)
 *   "f16_gram"     f16x3 mode: 1 (default) the two Gram products on the fp16 pipe too, 0 keep fp32 mode's
 *   "xtz_nsplit"   row splits of X~^T Zbar (0 = default)
 *   "use_graph"    0: scfgp_train launches every iteration eagerly instead of replaying a captured hipGraph (1, default: the
 *                  graph unless a communicator is attached; 2: the graph with a communicator too)
 *   "apply_dma"    the tiles of the apply products staged by LDS-DMA (global_load_lds) instead of through registers:
 *                  -1 automatic (K > 256 and >= 16384 rows: 128-wide tiles; fp32 from K >= 1024 and 65536 rows: 256-wide), 0 off,
 *                  1 = 128-wide tiles, 2 = 256-wide tiles (fp32; fp64 stays 128 wide)
 *   "gram64"       precision level policy of fp32 mode (scfgp_get_condition): 0 never, 1 always level 1, 2 auto, 3 always level 2
 *   "cond_threshold" / "cond_threshold_w"   the two thresholds of the auto policy
 *   "factor_form"  -1 auto (= precision level 2), 0 never, 1 always: pass 2 as C = Phi Li^T, V = C Li
 *   "lowrank_bwd"  -1 auto (D+1 >= 4 (S+1), padded), 0 never, 1 whenever it can (the forward projection goes through the S
 *                  columns and U fits): the reverse sweep of F = l_F r_F^T through T~^T Zbar and X~^T (Zbar_L + Zbar_M r_F)
 *                  instead of the dense X~^T Zbar; exchange buffer 3 then holds those two
 *   "roctx"        1: push a roctx range per stage for `rocprofv3 --marker-trace` (off by default; also SCFGP_ROCTX=1)
 *   "test_deny_level" / "test_fail_stage"   fault injection for the tests of "ranks decide together": precision levels >= value
 *                  are refused as if their buffers could not be allocated / the next sweep `value` (1..3) fails before it enqueues
 * Experiments of earlier rounds that measured equal or slower (feature map fused into the Gram loaders, lock-step Gram
 * schedule, pass 3 in row parts on two streams, Zbar written by the Phibar product, 8-wave and hand-pipelined LDS-DMA
 * tiles, the bf16x3 split-precision dtype) are no longer part of the library: profiles/r02_tuning.md, r03_tuning.md hold
 * their measurements, git history (tag of round 3: commit 757ac97) their code. */
int scfgp_set_option(scfgp_ctx* ctx, const char* name, int64_t value);

/* Box probe (no reference counterpart; bench.py's `secondary.box`): ~100 ms of device work on `device`, no context needed.
 * out[0] = fp32 MFMA TFLOP/s of a register-only v_mfma_f32_16x16x4_f32 loop (best of 1, 2, 8 waves per SIMD; n >= 6: each
 * in out[3..5]), out[1] = shader clock it held (GHz), out[2] = GB/s of a 1 GiB -> 1 GiB streaming copy (read + write), n >= 7:
 * out[6] = GB/s of a read-only stream over the same 2 GiB (the ceiling of the sweeps that only read).
 * Lets two timings from two devices be normalised. */
int scfgp_box_probe(int device, double* out, int n);

/* host-only self-test of the row splits of the Gram products (how the N rows are cut into the splits whose partial
 * sums the reduction adds; `nsplit` 0 = default, `taper` as the option): returns 0 when the splits tile [0, Np) in
 * order on 256-row blocks.  Needs no GPU. */
int scfgp_selftest_row_splits(int D, int S, int M, int64_t N, int dtype, int nsplit, int taper);

#ifdef __cplusplus
}
#endif
#endif /* SCFGP_HIP_H */
