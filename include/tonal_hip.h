/* tonal_hip.h - C ABI of libtonal_hip.so (gfx950 / MI355X).
 *
 * The reference (Daniel-Lin-S/decode_tonal_langauge) is pure Python and has no FFI; every
 * arithmetic call on its hot path goes to torch / scipy.  This header
 * is therefore the boundary *we* define: each entry point names the reference call site
 * (file:line under /root/reference) whose arithmetic it replaces.  The Python mirror of the
 * reference's classes (decode_tonal_langauge_amd/models/..., preprocess/signal/...) binds these
 * with ctypes (decode_tonal_langauge_amd/_lib.py); INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host;
 *  - the caller owns every buffer, including workspaces; no entry point allocates, frees or
 *    synchronises; work is enqueued on `stream` (a hipStream_t passed as void*);
 *  - return 0 on success, a negative TL_E* code otherwise; tl_last_error() gives the
 *    thread-local message of the last failure;
 *  - fp32 unless stated; "rows" are flattened (sequence, time) rows of channels-last
 *    activations: row = seq * Tp + t, seq = b * C + c (data layout).
 */
#ifndef TONAL_HIP_H
#define TONAL_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TL_OK 0
#define TL_EINVAL (-1)   /* bad argument / shape */
#define TL_ELAUNCH (-2)  /* hip launch failure   */
#define TL_ENODEV (-3)   /* no gfx950 device     */

const char* tl_last_error(void);
int tl_version(void);
/* number of visible HIP devices, or negative error (does not initialise a context) */
int tl_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Windowed NT GEMM  ("conv (k,1) as implicit GEMM", fp32 MFMA 32x32x2):
 *   acc[R][n] = sum_{j<J} sum_{k<K} Arow(R + j + row_shift)[k] * Bw[j][n][k]
 * Replaces nn.Conv2d((k,1)) + LeakyReLU + MaxPool2d((2,1)) forward and input-gradient
 * (models/synthesis_models.py:86-105,116-131), nn.Linear forward (:133-135), and the
 * h @ W_hh^T product of nn.LSTM (:112,164-166).
 *
 * loader: 0 DIRECT  A rows are read from `A` (row stride lda);
 *         1 UNPOOL  A rows are the un-pooled gradient dZ built on the fly from the pooled
 *                   gradient G (`A`, row stride lda) and the arg-max bits written by the
 *                   forward epilogue (`abits`, [prow][ld_abits] 32-bit words):
 *                   dZ[Rz][k] = G[Rz/2][k] if bit(Rz/2,k)==(Rz&1) and (Rz%Tp)<Tvalid_in else 0.
 * epilogue: 0 STORE  out[R][n] = acc (+bias[n]); with splitk>1 partial sums go to
 *                    out + z*slab_stride (no bias) and the caller reduces;
 *           1 LRELU  out[R][n] = lrelu(acc + bias[n]);
 *           2 POOL   out[R/2][n] = max over the row pair of lrelu(acc + bias[n]), 0 for rows
 *                    with (R%Tp) >= Tvalid; arg-max bit -> obits[R/2][n/32];
 *           3 MASK   out[R][n] = acc * (aux[R][n] > 0 ? 1 : slope)    (LeakyReLU backward)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* A; const uint32_t* abits; const float* Bw; const float* bias; const float* aux;
  float* out; uint32_t* obits;
  int64_t M;          /* output rows to compute                                   */
  int64_t A_rows;     /* rows addressable in A (G rows for UNPOOL)                */
  int N, K;           /* output columns, reduction length per tap (K % 4 == 0)    */
  int lda, ldb, ldo, ldaux, ld_abits, ld_obits;
  int J;              /* taps (1..7); Bw is [J][N][ldb]                           */
  int row_shift;      /* 0 forward, -(J-1) input-gradient                         */
  int Tp, Tvalid, Tvalid_in;
  float slope;
  int loader, epilogue;
  int splitk; int64_t slab_stride;
  int bm;             /* row-tile height: 256 (8 waves, large M), 128 (default) or 32 (skinny M) */
  /* 1 bit per element, rows x ceil(cols/32) words like abits/obits: the POOL epilogue also writes
   * "pooled output > 0" to osign (leading dimension ld_obits, may be null); the MASK epilogue reads
   * its LeakyReLU' mask from auxbits (leading dimension ld_auxbits) instead of the floats in aux
   * when auxbits is not null - 1/32 of the bytes                                                   */
  uint32_t* osign; const uint32_t* auxbits; int ld_auxbits;
  /* epilogue 4 (Winograd input-gradient kernels only): the masked result is G1 = dL/dZ of the first
   * conv stage (C_in = 1, models/synthesis_models.py:87-89) at its arg-max; instead of storing it
   * the epilogue contracts it with the raw signal: c1partial[row tile][j][col] = sum_rows G1[row][col]
   * * x[seq][2t + a + j] (j < c1kt taps) and [c1kt][col] = sum_rows G1 (bias), with a = arg-max bit
   * in c1bits (leading dimension ld_auxbits), x = c1x (S, c1T), rows valid for t < Tvalid.          */
  const float* c1x; const uint32_t* c1bits; float* c1partial; int c1T, c1kt;
  /* epilogue 5 (tl_conv3_wino43v_nt only, round 4): POOL that also writes V = the F(4,3) input transform of its pooled
   * OUTPUT for the next 3-tap stage: vout[vout_quads][6][ld_vout], quad Q' = pooled rows 4Q' .. 4Q'+5 of one sequence
   * (Tp % 8 == 0).  `out` becomes optional (null: the raw pooled rows are never stored).  The last quad of every
   * 512-row tile needs two rows of the next tile: it is left raw (rows 0..3 in its transform slots 0..3) and the tile
   * stores its own first two pooled rows to vhalo[tile][2][N]; tl_wino43_v_fixup finishes those quads.                */
  float* vout; float* vhalo; int64_t vout_quads; int ld_vout;
  /* tl_conv3_wino63v_nt, POOL epilogue: rows per sequence of out / obits / osign, [seq * out_tp + t'] (pooled rows t' >=
   * out_tp are dropped); 0: Tp / 2.  Decouples the row stride of a stage's output from the hex padding of its input.   */
  int out_tp;
  /* tl_conv3_wino63v_nt, epilogue 6 (MASKY): the input gradient of a stage hands the stage BELOW its operands instead of the
   * gradient rows G: vout = Y[hex][8][ld_vout] (Y = A dz: the second operand of that stage's weight gradient,
   * tl_conv3_wino63v_tn with loader 3) and vout2 = Vd[hex][8][ld_vout] (the operand of its input gradient), both in the pair
   * layout, hexes of the stage below (three output rows each; vout_quads of them); dz = G un-pooled with abits (ld_abits),
   * zero from Tvalid_in conv rows on; `out` is not written.  The first hex of every 768-row tile is finished by
   * tl_wino63_vd_fixup from vhalo[tile][2][N].                                                                        */
  float* vout2;
} tl_nt_params;
int tl_gemm_nt_window(const tl_nt_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Windowed TN GEMM (weight gradients, reduction over rows):
 *   slab[z][j*Mj + m][n] = sum_{R in split z} A[R + j][m] * Bz[R][n]
 * with Bz = B (DIRECT, rows masked by (R%Tp)<Tvalid) or the un-pooled dZ (UNPOOL, as above).
 * Replaces the weight-gradient of Conv2d/Linear/LSTM produced by loss.backward()
 * (models/synthesis_trainer.py:226).  The caller sums the `splitk` slabs.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* A; const float* B; const uint32_t* bbits; float* slab;
  int64_t Krows;      /* reduction rows                                            */
  int64_t A_rows, B_rows;
  int Mdim, Ndim;     /* A columns used (per tap), B columns                       */
  int lda, ldb, ldc, ld_bbits;
  int J;
  int Tp, Tvalid;
  int loader;
  int splitk; int64_t slab_stride;
  float* colsum;      /* optional.  tl_gemm_tn_window on its short-reduction kernel (Krows <= 512, splitk 1): colsum[m] = sum
                         over the rows of A[.][m], Mdim floats (the bias gradient of a Linear layer); tl_gemm_tn_window on its
                         one-tap direct kernel (J == 1, loader 0, Mdim > 32, Krows > 512): colsum[z][n] = sum over the valid
                         rows of split z of B[.][n], Ndim floats per split (the bias gradient of a 1x1 convolution, from the
                         launch that reads its output gradient anyway); tl_conv3_wino43v_tn / tl_conv3_wino63v_tn:
                         colsum[z][n] = sum over split z of the un-pooled
                         dZ column n (the bias gradient partial sums; Ndim floats per split), or null */
  float* vd; int ld_vd; /* optional (tl_conv3_wino43v_tn): also write Vd[quad][6][ld_vd] = the F(4,3) input transform of
                         the un-pooled dZ rows 4q-2 .. 4q+3 - the operand of the stage's input-gradient pass
                         (tl_conv3_wino43v_nt, MASK / conv1-weight-gradient epilogue); Krows / 4 quads, or null    */
  int part;           /* with vd: 0 both launches (the Vd-writing first C_in tile, then the other tiles), 1 / 2 only the
                         first / second of them (to put them on different streams)                                  */
  int bm;             /* tl_conv3_wino43v_tn: C_in tile, 0 = 128 when Mdim % 128 == 0 else 64; 64 / 128 force one   */
  int g_tp;           /* tl_conv3_wino63v_tn: rows per sequence of B / bbits, [seq * g_tp + t']; 0: Tp / 2           */
} tl_tn_params;
int tl_gemm_tn_window(const tl_tn_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Winograd F(4,3) on PRE-TRANSFORMED operands for the 3-tap convolutions that are followed by MaxPool (2,1)
 * (conv2, conv3: models/synthesis_models.py:91-97; their backward in loss.backward(),
 * models/synthesis_trainer.py:226): 6 channel contractions per 4 conv rows instead of 12.  The A/B partner of the
 * F(6,3) default below (TONAL_KERNELS wino=4); the in-loop-transform F(2,3) / F(4,3) generation of rounds 1-2 was
 * retired in round 6 (shapes neither V form covers run on the windowed GEMMs above).
 * Both the forward pass and the weight gradient of a stage consume the same V = B^T d of its input rows; on gfx950 the
 * fp32 MFMA shares the vector ALU's rate, so transform / staging instructions inside the GEMM kernels
 * displace matrix work one for one.  V[quad][6][ldv] (quad Q = input rows 4Q..4Q+5 of one sequence,
 * rows past the sequence taken as zero) is written once and the GEMM kernels are transform-free:
 *   tl_wino43_weights          w (O, I, 3, 1) -> forward taps fwd [6][O][ld_f] and input-gradient taps dgr [6][I][ld_d]
 *                              (flipped; either may be null)
 *   tl_wino43_input_transform  P (rows, C; whole sequences of Tp rows, Tp % 4 == 0) -> V (rows / 4 quads)
 *   tl_conv3_wino43v_nt        tl_gemm_nt_window's contract with J = 3, Bw = the 6 transformed taps [6][N][ldb], loader = 2:
 *                              A = V, lda = ldv, A_rows = quads held by V (>= M / 4); POOL / POOLV epilogue (forward) or
 *                              MASK / conv1-weight-gradient epilogue on Vd (input gradient, row_shift -2); K % 16 == 0;
 *                              M and Tp multiples of 4; both operands reach LDS by buffer_load .. lds
 *   tl_conv3_wino43v_tn        tl_gemm_tn_window's contract (UNPOOL) writing the 6 transform accumulators
 *                              slab[z][6][Mdim][ldc] (slab_stride >= 6*Mdim*ldc) with A = V (lda = ldv, A_rows = quads
 *                              held by V, a whole number of 8-quad K-steps: pad with zero quads); V by
 *                              LDS-DMA into a 4-slot ring, Y = A dy staged through registers; Mdim % 64 == 0
 *   tl_wino43_wgrad_finalize   red [6][I][ld] (slabs summed by the caller) -> dW (O, I, 3, 1)
 * tl_conv1_fwd_v (below) writes V of the first stage's output directly.
 * ------------------------------------------------------------------------------------------ */
int tl_wino43_weights(const float* w, float* fwd, float* dgr, int O, int I, int ld_f, int ld_d, void* stream);
int tl_wino43_wgrad_finalize(const float* red, float* gw, int O, int I, int ld, void* stream);
int tl_wino43_input_transform(const float* P, float* V, int64_t rows, int Tp, int C, int ldp, int ldv, void* stream);
int tl_conv3_wino43v_nt(const tl_nt_params* p, void* stream);
/* second half of epilogue 5: for every tile (64 output quads) of a tl_conv3_wino43v_nt launch that wrote V, the last quad
 * Q' = 64 t + 63 (if < quads): rows 0..3 from its slots 0..3, rows 4, 5 from vhalo[t + 1] - or zeros where the quad ends
 * its sequence (Tq = pooled rows per sequence) or is the last of the matrix - transformed in place.                     */
int tl_wino43_v_fixup(float* V, const float* vhalo, int64_t quads, int64_t tiles, int Tq, int C, int ldv, void* stream);
/* Vd[conv_rows / 4][6][ldv] of a pooled 3-tap stage from its output gradient G (g_rows pooled rows, ldg) and arg-max bits:
 * the F(4,3) input transform of the un-pooled dZ rows 4q-2 .. 4q+3 = the operand of tl_conv3_wino43v_nt with the MASK /
 * conv1-weight-gradient epilogue.  LDS-free and register-light: meant to run on a side stream beside the stage's
 * MFMA-bound weight-gradient kernel.                                                                                    */
int tl_wino43_unpool_transform(const float* G, const uint32_t* bits, float* V, int64_t conv_rows, int64_t g_rows, int Tp,
                               int Tvalid, int C, int ldg, int ld_bits, int ldv, void* stream);
int tl_conv3_wino43v_tn(const tl_tn_params* p, void* stream);
/* ------------------------------------------------------------------------------------------
 * Round 4 - Winograd F(6,3) on pre-transformed operands (csrc/tonal_wino63.hip): the same three passes of conv2 / conv3
 * (models/synthesis_models.py:91-97 and their backward in loss.backward(), models/synthesis_trainer.py:226) with HEXES:
 * 8 channel contractions per 6 conv rows (0.444 of the direct form's multiplies; F(4,3): 0.5).  Points 0, +-1, +-2,
 * +-1/2, inf (numerics: oracle/winograd_f63_gate.py).  A sequence holds Tp rows, Tp % 6 == 0 (% 12 where the pooled
 * output feeds another F(6,3) stage); V[hex][8][ldv], hex H of a sequence = its rows 6H .. 6H+7 (rows past the sequence
 * taken as zero), zero hexes appended to whole tiles (tl_wino63_nt_tile_rows() / 6 hexes; the callers of this package pad to
 * 128, a multiple of it); Vd the same of the un-pooled dZ rows 6H-2 .. 6H+5.
 *   tl_wino63_weights        w (O, I, 3, 1) -> forward taps [8][O][ld_f], input-gradient taps [8][I][ld_d] (flipped)
 *   tl_conv3_wino63v_nt      tl_conv3_wino43v_nt with A = V in hex form (loader 2, A_rows = hexes, whole tiles;
 *                            M % 6 == 0, K % 8 == 0, K >= 40, N % 32 == 0); epilogues POOL (+ out_tp), POOLV (vout in
 *                            hex form, vout_quads = hexes, Tp % 12 == 0), MASK, conv1-weight-gradient (4)
 *   tl_wino63_nt_tile_rows   conv rows of a row tile of tl_conv3_wino63v_nt (768: 128 hexes): vhalo holds 2 x C floats per row tile, c1partial one row per row tile, the fix-up passes
 *                            take tiles = ceil(M / that)
 *   tl_wino63_v_fixup        second half of POOLV: the last output hex of every row tile (rows 6, 7 from vhalo)
 *   tl_conv3_wino63v_tn      tl_conv3_wino43v_tn with hexes (Mdim % 128 == 0, Ndim % 64 == 0; slab[z][8][Mdim][ldc],
 *                            slab_stride >= 8*Mdim*ldc; B / bbits rows [seq * g_tp + t']); vd optional: Vd[hex][8][ld_vd].
 *                            loader 3: B = Y[hex][8][ldb] written by epilogue 6 of the stage above - no transform in the
 *                            kernel (Mdim % 256 == 0; colsum from Y plane 1)
 *   tl_wino63_wgrad_finalize red [8][I][ld] -> dW (O, I, 3, 1)
 *   tl_conv1_fwd_v6          tl_conv1_fwd_v writing V[S * Tp / 6][8][C1] (Tp % 6 == 0)
 * ------------------------------------------------------------------------------------------ */
int tl_wino63_weights(const float* w, float* fwd, float* dgr, int O, int I, int ld_f, int ld_d, void* stream);
int tl_conv3_wino63v_nt(const tl_nt_params* p, void* stream);
int tl_wino63_nt_tile_rows(void);
int tl_wino63_v_fixup(float* V, const float* vhalo, int64_t hexes, int64_t tiles, int Tq, int C, int ldv, void* stream);
int tl_conv3_wino63v_tn(const tl_tn_params* p, void* stream);
int tl_wino63_wgrad_finalize(const float* red, float* gw, int O, int I, int ld, void* stream);
/* second half of epilogue 6: the first hex of every row tile (tile rows / 3 hexes of the stage below) of a Vd written by it - its front row comes from
 * vhalo[tile - 1] (zero where the hex starts its sequence: hexes_per_seq)                                              */
int tl_wino63_vd_fixup(float* Vd, const float* vhalo, int64_t hexes, int64_t tiles, int hexes_per_seq, int C, int ldv, void* stream);
/* Y[conv_rows / 6][8][ldv] = A dz and Vd = B^T (dz rows 6h-2 .. 6h+5) of a pooled 3-tap stage from its output gradient G (rows
 * [seq * g_tp + t'], ldg) and arg-max bits, pair layout: the operands of tl_conv3_wino63v_tn (loader 3) and of the stage's input
 * gradient for a stage whose G comes from a kernel without epilogue 6 (HBM-bound; conv3 of the reference stack)             */
int tl_wino63_unpool_yvd(const float* G, const uint32_t* bits, float* Y, float* Vd, int64_t conv_rows, int64_t g_rows, int Tp,
                         int g_tp, int Tvalid, int C, int ldg, int ld_bits, int ldv, void* stream);
int tl_conv1_fwd_v6(const float* x, const float* w, const float* b, float* P, float* V, uint32_t* bits, uint32_t* sign,
                    int64_t S, int T, int ktaps, int C1, int Tp, int Tout, float slope, void* stream);
/* Round 5 - the CNN-RNN classifier's 7-tap (7,1) convolutions + LeakyReLU (models/deep_classifiers.py:250-256) on the F(6,3) NT
 * kernel: three 3-tap segments summed in the transform domain by ONE launch whose K loop runs over 3 K channels.
 *   tl_wino63_xform2     x rows [seq * Tp + t][C] -> V0 = B^T (rows 6h .. 6h+7), V1 = B^T (rows 6h+3 .. 6h+10), pair layout, rows
 *                        from Tvalid on taken as zero (segment 2 - rows 6h+6 .. 6h+13 - is V0 of hex h + 1: no third array)
 *   tl_wino63_weights7   w (O, I, taps = 7..9) -> taps [3 I / 8][8][O][8] (segment s = taps 3s .. 3s+2, missing ones zero)
 *   tl_conv7_wino63v_nt  A = V0, aux = V1 (A_rows hexes each, whole 128-hex tiles), Bw = the taps, K = I, ldb >= 3 K, J = 7..9,
 *                        loader 2, epilogue LRELU: out[R][n] = lrelu(conv + bias), R < M (M % 6 == 0).  Windows stop at the end
 *                        of their sequence EXCEPT in the last hex of a sequence (rows t >= Tp - 6), whose third segment is
 *                        the first hex of the next sequence: those rows are never valid outputs of a 7-tap convolution     */
int tl_wino63_xform2(const float* P, float* V0, float* V1, int64_t rows, int Tp, int Tvalid, int C, int ldp, int ldv, void* stream);
int tl_wino63_weights7(const float* w, float* fwd, int O, int I, int taps, void* stream);
int tl_conv7_wino63v_nt(const tl_nt_params* p, void* stream);
/* Round 5 - the input gradient of a ONE-tap pooled stage whose input is the pooled output of a 3-tap F(6,3) stage (conv4 of the
 * reference stack: nn.Conv2d(512, 256, (1,1)) + LeakyReLU + MaxPool behind conv3, models/synthesis_models.py:96-101, backward in
 * loss.backward(), models/synthesis_trainer.py:226) on the F(6,3) NT kernel: its eight batched GEMMs take the six rows of a hex
 * (same taps W^T in each; two batches idle), so the accumulators are the gradient rows G of the stage below and the MASKY epilogue
 * (epilogue 7 = epilogue 6 without the inverse transform) writes that stage's Y / Vd directly - tl_wino63_unpool_yvd and the
 * gradient rows themselves go.  Row geometry: [seq * Tp + t], Tp % 3 == 0 rows per sequence = the hexes of the stage below
 * (three of its pooled rows each); hex H here = rows 6 H .. 6 H + 5 (may straddle two sequences).
 *   tl_wino63_unpool_rows6     G (the stage's pooled output gradient, rows [seq * g_tp + t / 2], ldg) + its arg-max bits ->
 *                              A[ceil(rows / 6)][8][lda], pair layout: slot i < 6 = un-pooled row 6 H + i (zero from Tvalid on),
 *                              slots 6, 7 zero (written only when pad != 0: a zero-filled buffer stays valid)
 *   tl_wino63_weights1         w (O, I) -> taps [ldb / 8][8][I][8]: slot t < 6 = w[o][n], slots 6, 7 zero
 *   tl_conv1_wino63v_dgrad_nt  A as above (A_rows hexes, whole 128-hex tiles), Bw = the taps, M = sequences x Tp rows, N = I,
 *                              K = O (K % 8 == 0, K >= 40), J = 1, loader 2, epilogue 7; auxbits / abits = sign / arg-max words
 *                              of the stage below in the layout [seq * out_tp + t] (out_tp <= Tp), Tvalid_in its valid conv
 *                              rows; vout / vout2 / vhalo / vout_quads / ld_vout as epilogue 6; then tl_wino63_vd_fixup       */
int tl_wino63_unpool_rows6(const float* G, const uint32_t* bits, float* A, int64_t rows, int64_t g_rows, int Tp, int g_tp, int Tvalid,
                           int C, int ldg, int ld_bits, int lda, int pad, void* stream);
int tl_wino63_weights1(const float* w, float* taps, int O, int I, int ldb, void* stream);
int tl_conv1_wino63v_dgrad_nt(const tl_nt_params* p, void* stream);
/* ------------------------------------------------------------------------------------------
 * RCCL handle (round 5; SURVEY 8b names it): the exchange step of the data-parallel trainer - between loss.backward() and
 * optimizer.step(), models/synthesis_trainer.py:226-227 - on plain fp32 device buffers.  One communicator per process (= per GPU);
 * rank 0 draws the id, the host side broadcasts its 128 bytes out of band (the Python host: through the torch.distributed store).
 * librccl is resolved at first use (the copy the process has loaded, else librccl.so.1): the library loads without it.
 *   tl_comm_unique_id   id128 <- ncclGetUniqueId
 *   tl_comm_init        *comm <- ncclCommInitRank(nranks, id, rank) on the current HIP device; tl_comm_destroy frees it
 *   tl_allreduce        recv[i] = op over ranks of send[i], i < count (op 0 sum, 1 max, 2 min; send == recv allowed)
 *   tl_reduce_scatter   recv[recv_count] = this rank's block of the sum over ranks of send[nranks * recv_count]
 *   tl_all_gather       recv[nranks * send_count] = the ranks' send[send_count] in rank order
 * all asynchronous on `stream`.
 * ------------------------------------------------------------------------------------------ */
int tl_comm_unique_id(void* id128);
int tl_comm_init(void** comm, int rank, int nranks, const void* id128);
int tl_comm_destroy(void* comm);
int tl_allreduce(void* comm, const float* send, float* recv, int64_t count, int op, void* stream);
int tl_reduce_scatter(void* comm, const float* send, float* recv, int64_t recv_count, void* stream);
int tl_all_gather(void* comm, const float* send, float* recv, int64_t send_count, void* stream);
/* sizeof() of the two parameter structs as compiled into the library (binding self-check) */
int tl_sizeof_nt_params(void);
int tl_sizeof_tn_params(void);

/* ---- first conv stage, C_in = 1 (models/synthesis_models.py:87-89) ----------------------
 * x (S, T) -> P1 rows (S*Tp, C1) + arg-max bits (+ sign bits "P1 > 0", may be null); w (C1,3) b (C1). */
int tl_conv1_fwd(const float* x, const float* w, const float* b, float* P, uint32_t* bits, uint32_t* sign,
                 int64_t S, int T, int ktaps, int C1, int Tp, int Tout, float slope, void* stream);
/* the same stage writing V = the F(4,3) input transform of its pooled output (V[S * Tp / 4][6][C1], Tp % 4 == 0)
 * for tl_conv3_wino43v_nt / _tn of the next stage; P is optional (null: the raw pooled rows are never stored)     */
int tl_conv1_fwd_v(const float* x, const float* w, const float* b, float* P, float* V, uint32_t* bits, uint32_t* sign,
                   int64_t S, int T, int ktaps, int C1, int Tp, int Tout, float slope, void* stream);
/* its weight/bias gradient from G1 = dL/dZ at the arg-max: partial[nblk][(ktaps+1)*C1]      */
int tl_conv1_wgrad(const float* x, const float* G, const uint32_t* bits, float* partial,
                   int nblk, int64_t S, int T, int ktaps, int C1, int Tp, int Tout, void* stream);

/* nn.Dropout of the deep classifiers (models/deep_classifiers.py:81,258) in train mode, in place on a flat buffer:
 * x[i] = keep(seed, i) ? x[i] / (1 - p) : 0 with the counter-hash stream of tl_concat_pack (index = position i)   */
int tl_dropout_scale(float* x, int64_t n, float p, uint64_t seed, void* stream);

/* ---- small strided helpers --------------------------------------------------------------- */
/* dst[i0][i1][i2][i3] (contiguous) = sum_{z<nz} src[z*zs + i0*s0 + i1*s1 + i2*s2 + i3*s3];
 * elements whose i_d >= lim_d (source extents) read as 0. Used to pack torch-layout weights
 * into GEMM layouts and to reduce + unpack split-K weight-gradient slabs.                   */
int tl_permute_reduce(const float* src, float* dst, const int64_t dims[4], const int64_t strides[4],
                      const int64_t lims[4], int nz, int64_t zs, const float* bias_last, void* stream);
/* out[c] partial sums over valid rows: partial[nblk][ncols]; rows valid iff (r%Tp)<Tvalid   */
int tl_colsum(const float* G, float* partial, int nblk, int64_t rows, int ncols, int ld,
              int Tp, int Tvalid, void* stream);

/* ---- label LSTM (models/synthesis_models.py:112,164-167; :249-252,288-289) --------------- */
/* one time step, pointwise part: gates = hh + x_t W_ih^T + b_ih + b_hh -> (i,f,g,o), c, h.
 * hh (U, 4H) may be null for t = 0 (zero state).  act (U,4H) keeps the activated gates.     */
int tl_lstm_cell_fwd(const float* hh, const float* x_t, const float* w_ih, const float* b_ih,
                     const float* b_hh, const float* c_prev, float* act, float* c, float* h,
                     int U, int H, int in_dim, int ld_hh, void* stream);
/* dgates . W of the label LSTM on its U <= 8 distinct label rows as a pure stream of W (N x K, row stride ldw;
 * label_lstm.weight_hh_l0: 73 728 x 18 432 at the north-star shape) - torch's LSTM backward inside loss.backward(),
 * models/synthesis_trainer.py:226:  slab[b][u][k] = sum_{n in row block b} g[u][n] W[n][k]; ceil(N / rows_per_block) slabs of
 * U x K floats, summed by the caller (tl_permute_reduce); rows_per_block 8..1024; K % 4 == 0, 16-byte aligned W / slab.       */
int tl_lstm_gw(const float* g, const float* W, float* slab, int U, int64_t N, int K, int64_t ldg, int64_t ldw, int rows_per_block,
               void* stream);
/* inference-only sequence (models/deep_classifiers.py:230-233,294-296,316-318: the CNN-RNN classifier's two LSTMs, run
 * forward-only by the synthesis trainer): xp[u*xp_row_stride + t*4H + gate*H + k] = input projection + both biases of every
 * step, pre-computed by one GEMM (rows are (b, t): xp_row_stride = T*4H or more).
 * One fused launch per step (recurrent product + cell update; no split-K slabs): wp = the recurrent weight
 * packed unit-major (row 4 u + g = W_hh row g H + u), H % 8 == 0, h_a / h_b ping-pong state buffers (the
 * first step ignores both); *last_in_b tells which one holds h_T.                                        */
int tl_lstm_infer_seq_fused(const float* xp, int64_t xp_row_stride, const float* wp, float* h_a, float* h_b, float* c,
                            int B, int H, int T, int* last_in_b, void* stream);
/* backward of the same step: dh, dc_next -> dgates (U,4H) row-major and transposed (4H,ldt),
 * dc_prev.  c_prev may be null (t = 0).                                                     */
int tl_lstm_cell_bwd(const float* dh, const float* dh_rec, const float* dc_next, const float* act,
                     const float* c, const float* c_prev, float* dgates, float* dgates_t, float* dc_prev,
                     int U, int H, int ldt, void* stream);
/* dW_ih (4H,in_dim), db (4H) from dgates (L,U,4H) and x (L,U,in_dim)                        */
int tl_lstm_ih_grad(const float* dgates, const float* x, float* dw_ih, float* db, float* db2 /* optional second copy of db: bias_hh */,
                    int L, int U, int H, int in_dim, void* stream);

/* ---- concat / dropout glue (models/synthesis_models.py:107,160-170) ----------------------- */
/* Xc[row][0:Cc] = O5[row][0:Cc] * keep(row,ch); Xc[row][Cc+lc] = h[uid[b]][lc*lat*C + t*C + c];
 * remaining pad columns = 0.  keep = 1 if p_drop == 0 else Bernoulli via a counter hash of
 * (seed, (drop_row0 + row)*Cc + ch): drop_row0 = first row of this rank's shard in the GLOBAL
 * batch (b0*C*Tp), so a data-parallel run draws the masks of the single-process run.        */
int tl_concat_pack(const float* O5, const float* h, const int32_t* uid, float* Xc,
                   int B, int C, int Tp, int lat, int Cc, int Lc, int ld5, int ldh, int ldx,
                   float p_drop, uint64_t seed, int64_t drop_row0, void* stream);
/* backward: G5[row][ch] = dXc[row][ch]*keep*lrelu'(O5); dh[u][..] = sum_{b in u} dXc[..][Cc+lc], the members of u
 * taken from members[offsets[u] .. offsets[u+1]) - or, with members == NULL, by scanning offsets = the batch's label
 * ids (B entries) in batch order (the order a stable sort would give: same sums, nothing to prepare)              */
int tl_concat_unpack_bwd(const float* dXc, const float* O5, const int32_t* members, const int32_t* offsets,
                         float* G5, float* dh, int B, int U, int C, int Tp, int lat, int Cc, int Lc,
                         int ld5, int ldh, int ldx, float slope, float p_drop, uint64_t seed, int64_t drop_row0,
                         void* stream);

/* ---- loss / metric (models/synthesis_trainer.py:14-43,140,222-229) ------------------------ */
/* targets are truncated toward zero when trunc_targets != 0 (the reference's .long()).
 * dout (row stride ldd) = sign(out - t) / (B*D) * grad_scale; stats[0] += L1 mean,
 * stats[1] += MCD mean, stats[2], stats[3] = this call's L1 / MCD                            */
int tl_l1_mcd(const float* out, const float* targets, float* dout, float* stats,
              int B, int D, int ldd, int trunc_targets, float grad_scale, void* stream);

/* ---- NAdam (torch.optim.NAdam as built at models/synthesis_trainer.py:131-137) ------------ */
int tl_nadam(float* p, const float* g, float* m, float* v, int64_t n, float coef_grad, float coef_mom,
             float beta1, float beta2, float bias_corr2, float eps, float weight_decay, float grad_scale,
             void* stream);
/* One launch for a list of tensors sharing the step's scalars.  entries (DEVICE memory, count of them,
 * every pointer 16-byte aligned): block0 = first block of the tensor = sum over the tensors before it of
 * ceil(n / tl_nadam_multi_chunk()); total_blocks = that sum over all.  Same arithmetic as tl_nadam.   */
typedef struct { float* p; const float* g; float* m; float* v; int64_t n; int64_t block0; } tl_nadam_entry;
int tl_nadam_multi(const tl_nadam_entry* entries_dev, int count, int64_t total_blocks, float coef_grad, float coef_mom,
                   float beta1, float beta2, float bias_corr2, float eps, float weight_decay, float grad_scale,
                   void* stream);
int tl_nadam_multi_chunk(void);
/* tl_nadam_multi with the step's scalars (coef_grad, coef_mom, bias_corr2) read from device memory scalars_dev[0..2]: the
 * launch can be captured in a HIP graph and replayed with new values (models/synthesis_trainer.py:227, one optimizer.step
 * per batch)                                                                                                         */
/* scalars_dev[0..2] = (coef_grad, coef_mom, bias_corr2), *seed_dev = seed (either may be null): the values travel as
 * launch arguments, stream-ordered in front of a graph replay that reads them                                       */
int tl_set_step_scalars(float* scalars_dev, uint64_t* seed_dev, float coef_grad, float coef_mom, float bias_corr2,
                        uint64_t seed, void* stream);
/* dstA[i] = sum_{z < nz} srcA[z nA + i], dstB[j] = sum_z srcB[z nB + j] (nB may be 0): two slab reductions in one launch - the
 * per-window partials of a Conv1d's weight and bias gradient (SynthesisLite, models/synthesis_models.py:236-244)          */
int tl_sum_slabs2(const float* srcA, float* dstA, int64_t nA, const float* srcB, float* dstB, int64_t nB, int nz, void* stream);
/* tl_set_step_scalars plus the step's input tensors (n <= 4: host tables of device pointers and byte counts) copied into the
 * static buffers a captured step reads - all a replay needs in ONE launch (models/synthesis_trainer.py:201-205, the batch
 * tuple of the loop)                                                                                                    */
int tl_stage_step(float* scalars_dev, uint64_t* seed_dev, float coef_grad, float coef_mom, float bias_corr2, uint64_t seed,
                  const void* const* src, void* const* dst, const int64_t* nbytes, int n, void* stream);
int tl_nadam_multi_dev(const tl_nadam_entry* entries_dev, int count, int64_t total_blocks, const float* scalars_dev,
                       float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream);
/* the same update for a parameter (rows x cols) whose gradient is low rank, g = fa^T . fb with
 * fa (kr, rows), fb (kr, cols), kr <= 64: the gradient is formed in registers and never stored
 * (label_lstm.weight_hh_l0: 5.4 GB less written and read per step).                              */
int tl_nadam_lowrank(float* p, float* m, float* v, const float* fa, const float* fb, int kr, int rows, int cols,
                     int ldfa, int ldfb, float coef_grad, float coef_mom, float beta1, float beta2,
                     float bias_corr2, float eps, float weight_decay, float grad_scale, void* stream);
/* ... and, in the same pass over p, dh = fa[0:U] . p_OLD (U x cols): the last step of the label LSTM's BPTT
 * (dh_1 = dgates_2 . W_hh reads the weight as it was BEFORE this update, and the W_hh gradient has no term for the first step,
 * so the update does not wait for it; torch's LSTM backward inside loss.backward(), models/synthesis_trainer.py:226, followed by
 * optimizer.step(), :227).  Rows 0..U-1 of fa must be the dgates of that step (U <= 8, U <= kr).  The kernel writes
 * ceil(rows / (32 row_tiles)) partial slabs dh_slab[slab][U][cols] (caller-owned, 16-byte aligned); their sum over `slab`
 * (tl_permute_reduce) is dh.  One 5.4 GB stream of W_hh less per train step.                                                 */
int tl_nadam_lowrank_dh(float* p, float* m, float* v, const float* fa, const float* fb, int kr, int rows, int cols,
                        int ldfa, int ldfb, float coef_grad, float coef_mom, float beta1, float beta2,
                        float bias_corr2, float eps, float weight_decay, float grad_scale, float* dh_slab, int U,
                        int row_tiles, void* stream);

/* ---- tone dynamics gather (data_loading/utils.py:32-79) ----------------------------------- */
/* labels[b][0][l] = syl[b]; labels[b][1][l] = table[tone[b]][l]; err flag set if tone out of range */
int tl_tone_dynamics(const int64_t* tone, const int64_t* syl, const float* table, float* labels,
                     int32_t* err, int B, int n_tones, int L, void* stream);
/* out[i] = LeakyReLU(sum_z slab[z][i] + bias[i % ncols]), i < n: split-K reduction + bias + activation of a Linear layer
 * (models/synthesis_models.py:252-256, fc.1 of SynthesisLite)                                                       */
int tl_splitk_bias_lrelu(const float* slab, const float* bias, float* out, int nz, int64_t n, int ncols, float slope, void* stream);
/* out (B, N) = x (B, K; row stride ldx) . w (N, K)^T + bias (N, may be NULL): the Linear layer of LogisticRegressionClassifier
 * on the flattened window (models/simple_classifiers.py:34-60) and the output layer of ShallowNNClassifier (:112-121) - a
 * handful of columns over a long K, where a GEMM tile would be empty.  act = 1 applies the sigmoid the deep classifiers end
 * with (models/deep_classifiers.py:97-99,265-267).  N <= 64, K % 4 == 0, 16-byte aligned rows.                        */
int tl_linear_rows(const float* x, const float* w, const float* bias, float* out, int B, int K, int N, int64_t ldx, int act,
                   void* stream);
/* the whole label pass of a train step (models/synthesis_trainer.py:207-218) in one launch: tone = argmax of tone_scores
 * (B, n_tone_cls), syl = argmax of syl_scores (B, n_syl_cls) (first maximum), the gather of tl_tone_dynamics (n_rows table
 * rows) and, if pair is given, pair[b] = tone * n_syl + syl                                                        */
int tl_labels_from_scores(const float* tone_scores, const float* syl_scores, const float* table, float* labels, int64_t* tone,
                          int64_t* syl, int32_t* pair, int32_t* err, int B, int n_tone_cls, int n_syl_cls, int n_rows, int n_syl,
                          int L, void* stream);

/* ---- SynthesisLite blocks (models/synthesis_models.py:236-263,265-296) ----------------------
 * x (B,Cin,T) channels-first like the reference's Conv1d; 'same' padding (2*pad == k-1).       */
/* z = conv1d(x,w)+bias; part[(b*ntile+tile)][Cout][2] = per-tile (sum z, sum z^2), ntile=ceil(T/64) */
int tl_lite_conv_fwd(const float* x, const float* w, const float* bias, float* z, float* part,
                     int B, int Cin, int Cout, int T, int k, int pad, void* stream);
/* BatchNorm1d statistics: training -> batch mean / rstd from `part` (+ running-stat update,
 * unbiased variance); eval -> from the running buffers                                        */
int tl_lite_bn_finalize(const float* part, float* mean, float* rstd, float* run_mean, float* run_var,
                        int nparts, int C, int64_t count, float momentum, float eps, int training,
                        int64_t* tracked /* optional: BatchNorm's num_batches_tracked, += 1 in training */, void* stream);
/* y (B,C,T/2) = MaxPool1d(2)(LeakyReLU(BN(z)))                                                 */
int tl_lite_bn_act_pool_fwd(const float* z, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, float* y, int B, int C, int T, float slope, void* stream);
/* backward of the same: dy (B,C,T/2) -> dz (B,C,T), dgamma, dbeta; work holds (B*C*2 + C*2) floats */
int tl_lite_bn_act_pool_bwd(const float* dy, const float* z, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, float* dz, float* dgamma, float* dbeta,
                            float* work, int B, int C, int T, float slope, int training, void* stream);
/* conv backward: dx (B,Cin,T) (may be null), dwpart (B,Cout*Cin*k), dbpart (B,Cout) per-sample partials */
int tl_lite_conv_bwd(const float* dz, const float* x, const float* w, float* dx, float* dwpart, float* dbpart,
                     int B, int Cin, int Cout, int T, int k, int pad, void* stream);
/* label LSTM over the whole sequence: xl (B,L,in_dim) -> act (B,L,4H), cs, hs (B,L,H)          */
int tl_lite_lstm_fwd(const float* xl, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                     float* act, float* cs, float* hs, int B, int L, int H, int in_dim, void* stream);
/* BPTT from the gradient of the last hidden state: dgates (B,L,4H)                             */
int tl_lite_lstm_bwd(const float* dh_last, const float* w_hh, const float* act, const float* cs, float* dgates,
                     int B, int L, int H, int ld_dh, void* stream);
/* feat (B,ldf) = Dropout([flatten(y2) | h_L]) and its backward split                           */
int tl_lite_cat(const float* y2, const float* hs, float* feat, int B, int F, int H, int L, int ldf,
                float p_drop, uint64_t seed, void* stream);
int tl_lite_uncat(const float* dfeat, float* dy2, float* dh, int B, int F, int H, int ldf, float p_drop,
                  uint64_t seed, void* stream);
/* the same two with the dropout seed in device memory (*seed_dev): capturable in a HIP graph whose replays draw new masks */
int tl_lite_cat_dev(const float* y2, const float* hs, float* feat, int B, int F, int H, int L, int ldf,
                    float p_drop, const uint64_t* seed_dev, void* stream);
int tl_lite_uncat_dev(const float* dfeat, float* dy2, float* dh, int B, int F, int H, int ldf, float p_drop,
                      const uint64_t* seed_dev, void* stream);

/* ---- preprocess/signal band extraction (preprocess/signal/frequency_filter.py) ------------ */
/* Gaussian-bank analytic envelope, circular, exact DFT-domain taps supplied by the host:
 * taps (nb, ntap, 2) complex float64 kernels h_b[n], n = k - half for k in [0, ntap), ntap <= T
 * (frequency_filter.py:158-184); x (C,T) float32 or float64 (x_is_f64), y (C,T) float64 =
 * mean_b |sum_n h_b[n] x[(t-n) mod T]| (envelope != 0) or the mean of the real parts.        */
int tl_gauss_envelope(const void* x, int x_is_f64, const double* taps, double* y, int C, int64_t T,
                      int nb, int ntap, int half, int envelope, void* stream);
/* The same bank through the kernel's Hermitian symmetry (the reference's per-band DFT multiplier is real:
 * h_b[-n] = conj(h_b[n]), frequency_filter.py:155-175): taps (half + 1, 8, 2) = Re / Im of h_b[n], n = 0..half (tap-major); 0.56 x
 * the fp64 operations of tl_gauss_envelope.  8 bands, 2 half + 1 <= T taps within the LDS window.                   */
int tl_gauss_envelope_sym(const void* x, int x_is_f64, const double* taps, double* y, int C, int64_t T, int nb, int half,
                          int envelope, void* stream);
/* The same bank by overlap-save on an LDS-resident FFT (fp64) of nfft = 1024 points: G (8, nfft, 2) = FFT_nfft of each
 * band's truncated kernel (h_b[n], n = -half..half, placed at 0..2 half) / nfft, tw (nfft, 2) = (cos, -sin)(2 pi m / nfft);
 * 2 half <= nfft / 2.
 * 2.4 x fewer fp64 operations per sample than tl_gauss_envelope_sym at 209 taps.                                       */
int tl_hilbert_ols(const void* x, int x_is_f64, const double* G, const double* tw, double* y, int C, int64_t T, int nb,
                   int half, int nfft, int envelope, void* stream);
/* tl_hilbert_ols for band-limited kernels (the Gaussian bank: frequency_filter.py:155-175): when every |G_b| outside a window
 * of nfft / 4 bins [k0_b, k0_b + nfft / 4) is negligible (the caller checks: < 1e-10 of the peak in sum), band b's inverse
 * transform is four independent (nfft / 4)-point transforms, one wave each, with no workgroup barrier in the band loop.
 * Gp (nb, 4, nfft / 4, 2) = G_b[(k0_b + k) % nfft] . exp(+2 pi i k r / nfft), r < 4; k0 (nb) int32, device memory.
 * x_is_f64: 1 = float64 recording, fp64 transforms; 0 = float32 recording, fp32 transforms - the reference's own arithmetic
 * for that dtype (scipy.fft keeps single precision: complex64, frequency_filter.py:167-181); 2 = float32 recording, fp64
 * transforms (round 3's behaviour).  The output is float64 either way, as the reference returns it.                    */
int tl_hilbert_ols_bl(const void* x, int x_is_f64, const double* Gp, const int* k0, const double* tw, double* y, int C,
                      int64_t T, int nb, int half, int nfft, int envelope, void* stream);
/* The same bank evaluated in the DFT domain exactly as the reference writes it (frequency_filter.py:155-184):
 * X = DFT(x), z_b = IDFT(X . K_b), y = mean_b |z_b| (envelope) or mean_b Re z_b.  Arbitrary T (Bluestein chirp-z
 * over radix-2 Stockham passes, fp64).  kernels (nb, T) real = H_b x analytic multiplier; w (T,2) chirp, bf (m2,2) FFT
 * of the chirp filter, tw (m2/2,2) twiddles, m2 >= 2T-1 a power of two: coefficient data from the host, as for
 * tl_fft_resample.  work: (2 C m2 + C T) complex128.  No limit on the kernel length: used when the time-domain
 * taps exceed tl_gauss_envelope's LDS window (low bands at a raw recording rate).                              */
int tl_hilbert_fft(const void* x, int x_is_f64, double* y, int C, int64_t T, const double* kernels, int nb,
                   const double* w, const double* bf, const double* tw, int m2, int envelope, double* work, void* stream);
/* zero-phase IIR (scipy filtfilt, odd padding, lfilter_zi), fp64 (frequency_filter.py:226-227);
 * work holds 2*C*(T + 6*ntaps) doubles                                                        */
int tl_filtfilt_f64(const void* x, int x_is_f64, const double* b, const double* a, const double* zi,
                    double* y, double* work, int C, int64_t T, int ntaps, void* stream);
/* the same zero-phase filter evaluated TIME-PARALLEL (opt-in; frequency_filter.py:226-227): blocks of L samples run scipy's
 * recurrence from a zero state, the block-start states follow from a per-channel scan with the matrices
 * M[m] = A^(L 2^m) (nlev x 8 x 8 x (hi, lo) double-double pairs, from the host: exact powers of the recurrence's transition
 * matrix rounded once; nlev >= 7: the scan takes chunks of 128 blocks), and the blocks are re-run from their true start states.  Agrees with
 * tl_filtfilt_f64 to 2e-8 - 5e-8 relative - the size of the reference's own rounding error - not to the last bit.
 * ntaps <= 9, L >= 8, at most 65 535 blocks.  work: 2*C*(T + 6*ntaps) doubles; swork: 2 * blocks * 8 * C doubles.        */
int tl_filtfilt_scan_f64(const void* x, int x_is_f64, const double* b, const double* a, const double* zi, const double* M,
                         int nlev, double* y, double* work, double* swork, int C, int64_t T, int ntaps, int L, void* stream);
/* causal cascade of biquads (sosfilt), fp64 (frequency_filter.py:223-224)                     */
int tl_sosfilt_f64(const void* x, int x_is_f64, const double* sos, double* y, int C, int64_t T,
                   int nsec, void* stream);
/* causal FIR bank, mean over bands (frequency_filter.py:260-274); taps (nb, ntap) float64      */
int tl_fir_bank(const void* x, int x_is_f64, const double* taps, void* y, int y_is_f64, int C, int64_t T,
                int nb, int ntap, void* stream);
/* the same causal bank by overlap-save on the LDS-resident 1024-point FFT of tl_hilbert_ols: G (nb, 1024, 2) = FFT_1024 of each
 * band's taps / 1024, tw (1024, 2) the twiddle table; ntap <= 513                                                      */
int tl_fir_bank_ols(const void* x, int x_is_f64, const double* G, const double* tw, void* y, int y_is_f64, int C, int64_t T,
                    int nb, int ntap, void* stream);

/* ---- other preprocess/signal steps (same run(data, params) plugin ABI) --------------------------
 * per-channel z-score with statistics over [t0, t1): channel_zscore.py:22-27 (t0=0, t1=T) and
 * zscore_rereference.py:66-68 (baseline interval); y has the input dtype; stats (C,2) f64 workspace */
int tl_row_zscore(const void* x, int is_f64, void* y, double* stats, int C, int64_t T, int64_t t0, int64_t t1,
                  int zero_nans, void* stream);
/* common-average re-reference over the channels with include[c] != 0 (car_rereference.py:34-39)   */
int tl_car(const void* x, int is_f64, const int32_t* include, void* y, int C, int64_t T, int n_inc, void* stream);
/* pandas rolling(window, min_periods=1) z-score, sample std (rolling_zscore.py:36-49); y float64   */
int tl_rolling_zscore(const void* x, int is_f64, double* y, int C, int64_t T, int window, int zero_nans, void* stream);

/* FFT resampling = scipy.signal.resample(x, num, axis=1) (downsample.py:21-27): Bluestein chirp-z on
 * power-of-two FFTs.  Host-prepared coefficient arrays (complex128 interleaved): w1 (nx) / w2 (num)
 * chirps exp(i pi m^2 / n); bf1 (m2a) / bf2 (m2b) spectra of the chirp filters; tw1 (m2a/2) / tw2
 * (m2b/2) twiddles exp(-2 pi i k / m2).  work: C * (2*max(m2a,m2b) + m2b) complex128.  y has x's dtype. */
int tl_fft_resample(const void* x, int is_f64, void* y, int C, int64_t nx, int64_t num, const double* w1,
                    const double* bf1, const double* tw1, int m2a, const double* w2, const double* bf2,
                    const double* tw2, int m2b, double* work, void* stream);

#ifdef __cplusplus
}
#endif
#endif
