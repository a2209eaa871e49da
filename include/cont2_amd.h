/*
 * cont2_amd.h -- C-ABI of the MI355X-native contour-context hot path.
 *
 * This is the drop-in boundary of the build.  The reference has no
 * FFI layer: its interface is the C++ class API of the `cont2contops` library.  Every entry
 * point below names the reference function(s) it replaces (file:line under the reference
 * tree), and `contour-context_amd/hostcpp/` re-creates the reference classes on top of it.
 *
 * All functions return 0 on success, a negative CC_E* code otherwise; they never abort.
 * Pointers named d_* are device (HBM) pointers, h_* host pointers.  `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  No torch types cross this boundary.
 *
 * Layout structs (cc_contour_t, cc_bci_t, cc_scan_desc_t ...) are plain-old-data shared by the
 * device kernels, the host mirror and -- for comparison only -- the CPU oracle under oracle/.
 */
#ifndef CONT2_AMD_H
#define CONT2_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- compile-time shape ---- */
#define CC_NLEV 6          /* lv_grads_.size(); both shipped configs use 6 (yaml :30-31)        */
#define CC_KEY_DIM 10      /* RET_KEY_DIM, contour_mng.h:89                                     */
#define CC_NPIV 6          /* piv_firsts_ upper bound (anchors per level), contour_mng.h:107    */
#define CC_NDIST 10        /* dist_firsts_ upper bound, contour_mng.h:108                       */
#define CC_BCI_LAYERS 4    /* NUM_BIN_KEY_LAYER: DIST_BIN_LAYERS = {1,2,3,4}, contour_mng.h:113 */
#define CC_BITS_PER_LAYER 64 /* contour_mng.h:112                                               */
#define CC_BCI_MAXPTS 40   /* 4 layers x 10 neighbours                                          */
#define CC_MAXC 320        /* stored contours per level per scan (sorted, largest first)        */
#define CC_MAX_CELLS 22500 /* n_row_*n_col_ upper bound (150x150), LDS-resident BEV             */
#define CC_NQLEV 3         /* q_levels_.size() upper bound ([1,2,3], yaml :11)                  */
#define CC_KNN_MAX 64      /* nnk_ upper bound (shipped 50)                                     */
#define CC_GMM_LEVELS 4    /* GMMOptConfig::levels_ = {1,2,3,4}, correlation.h:18               */

/* Status codes.  Every entry point returns one; cc_last_error() (thread-local text) explains a non-zero one.
 * CC_EINVAL / CC_ECAPACITY leave the handle's state unchanged: it stays usable.  CC_EHIP means a HIP runtime
 * call failed part-way: destroy the handle (cc_db_destroy / cc_destroy release everything that was allocated).
 * Threading follows the reference (none): a cc_ctx and the cc_db objects created from it are to be
 * driven by one host thread at a time; different contexts (one per GPU / process) are independent. */
enum {
  CC_OK = 0,
  CC_EINVAL = -1,   /* bad argument / unsupported configuration            */
  CC_EHIP = -2,     /* a HIP runtime call failed (see cc_last_error)       */
  CC_ENOMEM = -3,
  CC_ECAPACITY = -4 /* a fixed capacity (DB size, batch size) was exceeded */
};

/* ------------------------------------------------------------------------- configs ------ */
/* ContourManagerConfig (contour_mng.h:92-110) + ContourViewStatConfig (contour.h:32-37). */
typedef struct {
  float lv_grads[CC_NLEV];
  float reso_row, reso_col;
  int32_t n_row, n_col;
  float lidar_height;
  float blind_sq;
  int32_t min_cont_key_cnt;
  int32_t min_cont_cell_cnt;
  int32_t piv_firsts;
  int32_t dist_firsts;
  float roi_radius;
  /* ContourViewStatConfig */
  int32_t min_cell_cov;
  float point_sigma;
  float com_bias_thres;
} cc_manager_cfg_t;

/* ContourSimThresConfig (contour.h:40-45). */
typedef struct {
  float ta_cell_cnt, tp_cell_cnt, tp_eigval, ta_h_bar, ta_rcom, tp_rcom;
} cc_sim_cfg_t;

/* CandidateScoreEnsemble (contour_db.h:244-250) = the three score unions
 * (contour_mng.h:121-219) flattened. */
typedef struct {
  int32_t i_ovlp_sum, i_ovlp_max_one, i_in_ang_rng; /* ScoreConstellSim */
  int32_t i_indiv_sim, i_orie_sim;                  /* ScorePairwiseSim */
  float correlation, area_perc, neg_est_dist;       /* ScorePostProc    */
} cc_score_t;

/* ContourDBConfig (contour_db.h:658-669) + TreeBucketConfig (:54-57). */
typedef struct {
  int32_t nnk;
  int32_t max_fine_opt;
  int32_t n_q_levels;
  int32_t q_levels[CC_NQLEV];
  cc_sim_cfg_t cont_sim;
  double max_elapse, min_elapse;
} cc_db_cfg_t;

void cc_default_manager_cfg(cc_manager_cfg_t *cfg); /* shipped KITTI values, yaml :27-47 */
void cc_default_db_cfg(cc_db_cfg_t *cfg);           /* yaml :6-23                        */
void cc_default_thresholds(cc_score_t *lb, cc_score_t *ub); /* yaml :69-87              */

/* ------------------------------------------------------------------ per-scan records ---- */
/* ContourView (contour.h:97-119).  eig_vecs is column-major like Eigen: [v00 v10 v01 v11],
 * column 1 (v01,v11) belongs to the larger eigenvalue. pos_cov likewise column-major. */
typedef struct {
  int16_t level;
  int16_t poi[2];
  int16_t cell_cnt;
  float pos_mean[2];
  float pos_cov[4];
  float eig_vals[2];
  float eig_vecs[4];
  float eccen;
  float vol3_mean;
  float com[2];
  uint8_t ecc_feat;
  uint8_t com_feat;
  uint8_t pad_[2];
} cc_contour_t; /* 76 bytes */

/* BCI::RelativePoint (contour_mng.h:245-258). */
typedef struct {
  int8_t level;
  int8_t seq;
  int16_t bit_pos;
  float r;
  float theta;
} cc_relpt_t; /* 12 bytes */

/* BCI (contour_mng.h:243-281).  dist_bin word w bit b  <=>  std::bitset bit 64*w+b.
 * n_segs = nei_idx_segs_.size() (0 when nei_pts_ is empty, else #distinct bit_pos + 1). */
typedef struct {
  uint64_t dist_bin[CC_BCI_LAYERS];
  int8_t piv_seq;
  int8_t level;
  uint8_t n_pts;
  uint8_t n_segs;
  uint16_t segs[CC_BCI_MAXPTS + 2];
  cc_relpt_t pts[CC_BCI_MAXPTS];
} cc_bci_t; /* 32 + 4 + 84 + 480 = 600 bytes */

/* Everything ContourManager keeps after makeContoursRecurs() + clearImage()
 * (contour_mng.h:426-436): sorted contour tables, per-level totals, 36 keys, 36 BCIs.
 * cont_perc_[l][j] is not stored: it is cell_cnt * 1.0f / layer_cell_cnt[l]
 * (contour_mng.h:607) and is recomputed bit-identically where needed. */
/* Descriptor flags.  The reference has no capacities; this build STORES at most CC_MAXC contours per level (only the
 * first piv_firsts_/dist_firsts_ and the largest ~95 % of the area are ever read downstream) and handles any number:
 * a scan with more than CC_MAXC components on a level is redone by an exact slow path (cc_k_contours_big: up to 3 840
 * per level, more than a 150 x 150 image can hold) behind the same call.
 *   CC_DESC_TRUNCATED          : a level has more than CC_MAXC contours, the CC_MAXC largest are stored in std::sort's
 *                                order; everything in the descriptor is exact (the oracle sets the same bit)
 *   CC_DESC_INEXACT_COMPONENTS : not produced any more (rounds 1-4: more than CC_MAXC components on a level); it stands
 *                                in a descriptor only between the fast launch and the slow one of the same call
 *   CC_DESC_INEXACT_KEYS       : a retrieval-key RoI held more cells than the kernel's list (roi_radius_ > 10 on a
 *                                densely built image): the keys are NOT exact
 * cc_ingest_host returns CC_ECAPACITY for CC_DESC_INEXACT_*; callers of cc_ingest_batch (device output) check `flags`
 * themselves.  KITTI/MulRan-like scans have tens to ~150 contours per level. */
#define CC_DESC_TRUNCATED 1
#define CC_DESC_INEXACT_COMPONENTS 2
#define CC_DESC_INEXACT_KEYS 4
typedef struct {
  int32_t n_cont[CC_NLEV];         /* cont_views_[l].size() (true count)                  */
  int32_t n_stored[CC_NLEV];       /* min(n_cont, CC_MAXC)                                */
  int32_t layer_cell_cnt[CC_NLEV]; /* layer_cell_cnt_                                     */
  float max_bin_val, min_bin_val;  /* contour_mng.h:436,524-525                           */
  int32_t n_pix;                   /* bev_pixfs_.size()                                   */
  int32_t flags;                   /* CC_DESC_* bits below; 0 = exact                     */
  float keys[CC_NLEV][CC_NPIV][CC_KEY_DIM];
  cc_bci_t bcis[CC_NLEV][CC_NPIV];
  cc_contour_t cont[CC_NLEV][CC_MAXC];
} cc_scan_desc_t;

/* The part of a scan's descriptor that the query path reads of a DATABASE scan (and of the query scan itself):
 * retrieval keys, the first dist_firsts_ contours and the BCIs of levels 1..4 (DIST_BIN_LAYERS = q_levels_' range =
 * GMMOptConfig::levels_, contour_mng.h:113, correlation.h:18).  18 KB instead of 169 KB: this is what the DB keeps
 * resident per scan (next to the correlation inputs, cc_gmm_feat), and -- together with them -- the wire format of the
 * multi-GPU exchange (cc_pack_scans / cc_db_add_packed).  hot.X[l] = desc.X[l + 1]. */
#define CC_HOT_LEVELS 4
typedef struct {
  int32_t n_cont[CC_HOT_LEVELS];
  int32_t layer_cell_cnt[CC_HOT_LEVELS];
  int32_t flags;
  int32_t pad_[3];
  float keys[CC_HOT_LEVELS][CC_NPIV][CC_KEY_DIM];
  cc_contour_t cont[CC_HOT_LEVELS][CC_NDIST]; /* rows >= n_stored are zero */
  cc_bci_t bcis[CC_HOT_LEVELS][CC_NPIV];
} cc_hot_desc_t; /* 48 + 960 + 3040 + 14400 = 18448 bytes */

/* Optional parity/debug outputs of ingest (device pointers, any may be NULL):
 *   bev     [n_scans][n_row*n_col] f32  : bev_ image (contour_mng.h:432), -1000 = empty
 *   pix_rc  [n_scans][n_row*n_col][2] f32: continuous (row_f,col_f) of the arg-max point of
 *                                          each occupied cell (Pixelf, contour_mng.h:392-411)
 *   labels  [n_scans][CC_NLEV][n_row*n_col] i16: canonical label image L_l(r,c) = seq of the
 *                                          owning contour after the size sort, -1 = none
 *                                          (SURVEY.md 8(a) "integer contour labels bit-exact")
 * Asking for bev or pix_rc makes the rasteriser write its dense arrays for the call's scans; without them it hands the
 * contour kernel the list of active cells only (and writes the dense arrays just for scans whose active cells overflow
 * that list). */
typedef struct {
  float *d_bev;
  float *d_pix_rc;
  int16_t *d_labels;
} cc_ingest_debug_t;

/* --------------------------------------------------------------------- query records ---- */
/* One KNN hit: IndexOfKey (contour_db.h:59-65) + squared distance. */
typedef struct {
  int32_t gidx; /* index of the scan in the DB (all_bevs_ position)  */
  int16_t level;
  int16_t seq;
  float dist_sq;
} cc_knn_hit_t;

/* Result of one query scan = what queryRangedKNN returns (contour_db.h:698-811) plus the
 * integer gate scores the parity contract compares. */
typedef struct {
  int32_t n_res;        /* 0 or 1 (CHECK(ptr_cands.size() < 2), batch_bin_test.cpp:187)      */
  int32_t cand_gidx;    /* DB index of the matched scan, -1 if none                          */
  double correlation;   /* cand_corr[0]                                                      */
  double tf[3];         /* T_delta = (x, y, theta) in BEV pixel units / radians              */
  int32_t cand_aft_check1, cand_aft_check2, cand_aft_check3; /* contour_db.h:357-359         */
  int32_t n_cand_pose;  /* candidates_.size() before tidyUpCandidates                        */
  int32_t n_cand_tidy;  /* candidates_.size() after tidyUpCandidates                         */
  int32_t n_knn_hits;   /* total KNN results over all query keys                             */
  int32_t flags;        /* CC_QF_* bits; 0 = exact.  Non-zero: an internal capacity was hit while this query was
                           scored (the reference has none), the result may differ from the reference's; the call
                           that collects the query returns CC_ECAPACITY (all results are still delivered)        */
  int32_t pad_;
} cc_query_result_t;
#define CC_QF_CHECK_CAP 1 /* a constellation check had more than 256 potential neighbour pairs (contour_mng.h:311-336)
                             or a rotation window of more than 63 pairs (:344-366): pairs were dropped              */
#define CC_QF_GMM_CAP 2   /* a scan of a correlation problem needs more ellipses of a level than the record holds (CC_MAXC; correlation.h:55-78) */
#define CC_QF_DESC_CAP 4  /* the correlation needed a contour beyond the CC_MAXC stored per level                     */
#define CC_QF_QUERY_INEXACT 8 /* the query scan's own descriptor is flagged CC_DESC_INEXACT_* (it exceeded a capacity of the
                                 contour kernel at ingest)                                                             */

/* ------------------------------------------------------------------------ context ------- */
typedef struct cc_ctx cc_ctx; /* opaque: device id, configs, scratch, streams */
typedef struct cc_db cc_db;   /* opaque: device-resident scan descriptors + key matrices +
                                 the host-side LayerDB bookkeeping                           */

const char *cc_last_error(void);
int cc_version(void);

/* Optional: start the device runtime, load the code object and put n_streams (0..32) streams into the per-device pool that
 * contexts and databases take theirs from.  Measured on MI355X / ROCm 7.2: first HIP call ~54 ms, code object 5-20 ms, a
 * stream 3.4-16 ms (a per-scan driver's context + database use 4-7).  A host that calls this when it starts (the class
 * mirror's ContourDB and evaluator constructors do) keeps those one-off costs out of its first scan.  Not calling it
 * changes nothing but when they are paid. */
int cc_runtime_init(int device, int n_streams);

/* Replaces ContourManager::ContourManager (contour_mng.h:478-498): validates the config
 * (n_row,n_col even, <= 150x150, 6 increasing levels) and allocates per-device scratch. */
int cc_create(int device, const cc_manager_cfg_t *cfg, int max_batch_scans, cc_ctx **out);
int cc_destroy(cc_ctx *ctx);

/* Per-kernel timing with HIP events recorded on the launch stream (bench.py roofline figures).
 * enable: subsequent cc_ingest_batch calls bracket K1 (rasterise) and K2 (contours) with events.
 * read  : synchronises, returns accumulated milliseconds {K1, K2} and the number of launches of
 *         each since the last read, then resets the accumulators. */
int cc_profile_enable(cc_ctx *ctx, int on);
int cc_profile_read(cc_ctx *ctx, double ms_out[2], int *n_launches);

/* ------------------------------------------------------------------------- ingest ------- */
/* Replaces, for a batch of scans, readKITTIPointCloudBin's point stream
 * (tools/pointcloud_util.h:9-47) -> ContourManager::makeBEV (contour_mng.h:505-556) ->
 * makeContoursRecurs (contour_mng.h:588-960, src/cont2/contour_mng.cpp:274-353).
 *   d_xyzi     : [total_points][4] f32 KITTI layout (x,y,z,intensity), device memory
 *   h_offsets  : [n_scans+1] point offsets of each scan into d_xyzi (host memory)
 *   d_out      : [n_scans] cc_scan_desc_t, device memory
 * Scans with <= 10 points violate CHECK_GT(size,10) (contour_mng.h:507) -> CC_EINVAL. */
int cc_ingest_batch(cc_ctx *ctx, const float *d_xyzi, const int64_t *h_offsets, int n_scans,
                    cc_scan_desc_t *d_out, const cc_ingest_debug_t *dbg, void *stream);

/* Same, from a host buffer (one H2D copy, then the device path); result copied back to
 * h_out.  This is what the ContourManager host mirror calls for a single scan. */
int cc_ingest_host(cc_ctx *ctx, const float *h_xyzi, const int64_t *h_offsets, int n_scans,
                   cc_scan_desc_t *h_out);
/* Same, and the max-height images too: h_bev [n_scans][n_row*n_col] f32 (bev_, contour_mng.h:432; -1000 = empty cell) --
 * what ContourManager::getBevImage / getContourImage / saveContourImage / saveMatchedPairImg read
 * (contour_mng.h:573-586, 1039-1049, 1286-1311; the SAVE_MID_FILE artefacts of the drivers).  h_bev may be NULL. */
int cc_ingest_host_bev(cc_ctx *ctx, const float *h_xyzi, const int64_t *h_offsets, int n_scans,
                       cc_scan_desc_t *h_out, float *h_bev);

/* ---- the per-scan loop (test/batch_bin_test.cpp:131-237 at sensor rate) ----
 * A cc_scan is ONE scan's descriptor kept on the device between ContourManager::makeContoursRecurs (contour_mng.h:588),
 * ContourDB::queryRangedKNN (contour_db.h:698) and ContourDB::addScan (:814): the class mirror's ContourManager holds one.
 * Nothing is allocated per scan: the context owns pinned staging buffers for the points, a device point buffer and a pool
 * of descriptor slots; the calls only queue work, the host copy of the descriptor (and of the max-height image, if asked
 * for) is fetched when a getter needs it.  Streams: cc_scan_ingest queues on the next of the context's two ingest CHANNELS
 * (own stream, device point buffer and one-scan scratch: consecutive scans' ingests overlap) and records the scan's `ready`
 * event behind its last kernel; cc_scan_desc / cc_scan_offload / cc_db_query_scan / cc_db_add_scan work on
 * the loop stream, which waits for `ready` first.  So scan i + 1 (and i + 2) can be ingested while scan i is queried and
 * added -- also from OTHER host threads: cc_ingest_batch / cc_stage_points* / cc_scan_ingest of one context serialise on
 * the context's ingest lock (the staging slots, the device point buffer and the K1/K2 scratch are shared), next to one
 * thread in the other cc_scan_* / cc_db_* calls (the slot pool is locked, cc_last_error is per thread).  The class
 * mirror's evaluator does exactly that (hostcpp/eval/evaluator.h: a helper thread reads the next files and ingests them).
 *   cc_stage_points  : pinned buffer for n_points x (x,y,z,i) f32 (write the points there to save a host copy), NULL on
 *                      failure.  The buffer belongs to the CALLING THREAD until that thread passes it to cc_scan_ingest
 *                      (or gives it up: cc_stage_points_cancel, or stages the same slot again); another thread that asks for
 *                      the same slot meanwhile waits.  cc_stage_points uses a slot of its own; cc_stage_points_slot names
 *                      slot 0 .. 2 * CC_SCAN_BATCH_MAX - 1 (each allocated when first asked for) -- so that the next scan's file can be read into one while the other's
 *                      scan is on its way to the device (a read-ahead thread alternates them, the driver thread's
 *                      cc_stage_points never collides with it); a slot is handed out again once ITS last copy has passed
 *                      (readKITTIPointCloudBin of scan i+1 next to queryRangedKNN of scan i, tools/pointcloud_util.h:9-47,
 *                      evaluator.h:285-302).  Asking for more points than the buffers hold re-allocates ALL slots (after
 *                      the ingest stream has drained; CC_EINVAL while another thread holds one)
 *   cc_stage_points_cancel : give a staged buffer back without ingesting it (short file, read error)
 *   cc_scan_ingest   : makeBEV + makeContoursRecurs for the points at h_xyzi (may be a staging pointer); want_bev != 0
 *                      keeps the max-height image for cc_scan_bev.  Returns at once (work is queued on the ingest stream).
 *   cc_scan_desc     : host copy of the descriptor (first call: one D2H copy + sync); CC_ECAPACITY if the scan exceeded a
 *                      capacity of the contour kernel (flags CC_DESC_INEXACT_*), the copy is delivered all the same
 *   cc_scan_offload  : move the descriptor to the host and give the device slot back; cc_scan_desc keeps working.  (The
 *                      mirror keeps the descriptors of the last CC_SCANS_ON_DEVICE = 8 192 added scans on the device --
 *                      169 KB each -- and offloads the oldest beyond that: a copy + sync per scan the loop does not need)
 *   cc_scan_release  : free the handle (waits for the scan's ingest and for queued readers of its slot) */
#define CC_SCAN_BATCH_MAX 16 /* scans per cc_scan_ingest_batch / cc_db_add_scan_batch / cc_db_query_scan_batch_submit */
typedef struct cc_scan cc_scan;
float *cc_stage_points(cc_ctx *ctx, int64_t n_points);
float *cc_stage_points_slot(cc_ctx *ctx, int64_t n_points, int slot);
int cc_stage_points_cancel(cc_ctx *ctx, const float *staged);
int cc_scan_ingest(cc_ctx *ctx, const float *h_xyzi, int64_t n_points, int want_bev, cc_scan **out);
/* cc_scan_ingest (want_bev = 0) for 1..CC_SCAN_BATCH_MAX scans at once: h_xyzi[i] must be staging buffers (cc_stage_points_slot,
 * distinct slots) in the calling thread's hands; ONE K1 / K2 launch chain for the batch (a chain takes ~0.2 ms of launch
 * latencies whatever it holds), out[i] are ordinary scan handles.  All or nothing: on an error no handle is returned.
 * cc_scan_ready: 1 once the scan's ingest has finished on the device, 0 while it is in flight (never blocks). */
int cc_scan_ingest_batch(cc_ctx *ctx, const float *const *h_xyzi, const int64_t *n_points, int n, cc_scan **out);
int cc_scan_ready(const cc_scan *scan);
int cc_scan_desc(cc_scan *scan, const cc_scan_desc_t **h_desc);
int cc_scan_bev(cc_scan *scan, const float **h_bev);
int cc_scan_offload(cc_scan *scan);
int cc_scan_on_device(const cc_scan *scan); /* 1: the descriptor still sits in a device slot, 0: offloaded (or NULL) */
int cc_scan_release(cc_scan *scan);

/* ---------------------------------------------------------------------- database -------- */
/* Replaces ContourDB::ContourDB (contour_db.h:680-684). */
int cc_db_create(cc_ctx *ctx, const cc_db_cfg_t *cfg, int capacity_scans, cc_db **out);
int cc_db_destroy(cc_db *db);
int cc_db_size(const cc_db *db);

/* Replaces ContourDB::addScan + ContourDB::pushAndBalance (contour_db.h:814-843) and
 * LayerDB::rebuild (src/cont2/contour_db.cpp:63-317) for n consecutive scans:
 * for i in [0,n): addScan(desc[i], h_ts[i]); pushAndBalance(h_seed[i], h_ts[i]).
 * The descriptors are appended to the device-resident DB; the bucket bookkeeping (which key
 * is searchable from which epoch on, bucket ranges per epoch) runs on the host.
 * Epoch e = state after e scans have been added and balanced.
 * An append does NOT wait for query chunks in flight (cc_db_query_submit): they were submitted against an earlier epoch
 * and keep reading the state they were submitted with (the sorted key view is double-buffered and a buffer is rewritten
 * only after its readers have finished -- a device-side wait; everything else is append-only).  So the online loop
 * ingest -> add -> submit(query at its own epoch) streams batch after batch without draining the GPU.  The call returns
 * once the host bookkeeping is done and its device work (key upload, sorted-view merge) is QUEUED on `stream`: queries
 * submitted afterwards wait for it on the device, whatever stream they come from; d_desc may be overwritten by work
 * queued on `stream` after the call. */
int cc_db_add_scans(cc_db *db, const cc_scan_desc_t *d_desc, int n, const double *h_ts,
                    const int32_t *h_seed, void *stream);

/* Optional first half of cc_db_add_scans for callers that stream batch after batch (the online loop of bench.py
 * --workload seq): queues, on `stream` and without waiting, what an append needs from the device before the host
 * bookkeeping can run -- the batch's compact records packed into the rows they will occupy, the retrieval keys
 * (key dimension 0 decides the bucket, contour_db.h:184-192) extracted and copied to pinned host memory.  Issued right
 * behind the batch's ingest, the copy travels while the previous batch is being queried; cc_db_add_scans on the same
 * (d_desc, n) then finds the keys on the host instead of waiting for a round trip.  The database is not changed.
 * At most two batches (<= 4096 scans each) may be prepared ahead; they must be added in the order they were prepared and
 * before anything else is added (CC_EINVAL otherwise); if the add of a prepared batch fails (CC_ECAPACITY: a scan flagged
 * inexact), a batch prepared behind it is dropped with it.  d_desc must not be overwritten before the add. */
int cc_db_add_scans_prepare(cc_db *db, const cc_scan_desc_t *d_desc, int n, void *stream);

/* Replaces ContourDB::queryRangedKNN (contour_db.h:698-811) for a batch of query scans.
 * Query i is answered against DB epoch h_epoch[i] (use cc_db_size() for "now"); in the
 * reference loop scan i queries epoch i (batch_bin_test.cpp:179 runs before :234-237).
 *   d_qdesc : [nq] query descriptors (device)
 *   h_res   : [nq] results (host)
 *   d_knn   : optional [nq][CC_NQLEV][CC_NPIV][CC_KNN_MAX] hits + d_knn_cnt [nq][3][6] i32
 *             (parity/debug; NULL to skip)
 *   thres_lb: the bars of the four gates and of the post-checks (CandidateScoreEnsemble sim_lb, contour_db.h:374-596)
 *   thres_ub: validated like CandidateManager's ctor does (lb.strictSmaller(ub), contour_db.h:365-367: CC_EINVAL otherwise)
 *             and otherwise UNUSED: the bars stay constant during a query, which is the reference's shipped DYNAMIC_THRES=0
 *             build (CMakeLists.txt:13-21).  The DYNAMIC_THRES=1 variant (contour_db.h:439-466, 566-574 raise the bars from
 *             candidate to candidate, a sequential dependence between a query's checks) is NOT implemented; the class mirror
 *             refuses to compile with that macro set (hostcpp/cont2/contour_db.h). */
int cc_db_query_batch(cc_db *db, const cc_scan_desc_t *d_qdesc, int nq, const int32_t *h_epoch,
                      const cc_score_t *thres_lb, const cc_score_t *thres_ub,
                      cc_query_result_t *h_res, cc_knn_hit_t *d_knn, int32_t *d_knn_cnt,
                      void *stream);

/* The same call split in two, for callers that stream batch after batch (offline evaluation of whole sequences,
 * tools/batch_eval.py, bench.py): cc_db_query_submit queues the batch's launch chains and returns; a chunk's results
 * reach h_res when its lane is needed again (a later submit) or at cc_db_query_wait, so the tail of one batch's chains
 * runs next to the head of the next batch's.  h_res must stay valid until cc_db_query_wait returned; d_qdesc may be
 * overwritten by work queued on `stream` after the call.  An error of an earlier batch's chunk (capacity flags) is
 * reported by the call that collects it.  cc_db_query_batch == submit + wait; cc_db_check_hints, cc_db_set_lanes and
 * cc_db_destroy collect the chunks in flight first; the appends do not need to (see cc_db_add_scans). */
int cc_db_query_submit(cc_db *db, const cc_scan_desc_t *d_qdesc, int nq, const int32_t *h_epoch,
                       const cc_score_t *thres_lb, const cc_score_t *thres_ub,
                       cc_query_result_t *h_res, cc_knn_hit_t *d_knn, int32_t *d_knn_cnt,
                       void *stream);
int cc_db_query_wait(cc_db *db);

/* Host-descriptor variants (one H2D copy each) used by the C++ class mirror, where a ContourManager owns a
 * host copy of its descriptor: ContourDB::addScan + pushAndBalance for one scan, and queryRangedKNN for one
 * query against the current DB state. */
int cc_db_add_scan_host(cc_db *db, const cc_scan_desc_t *h_desc, double ts, int32_t seed);
int cc_db_query_host(cc_db *db, const cc_scan_desc_t *h_qdesc, const cc_score_t *thres_lb, const cc_score_t *thres_ub,
                     cc_query_result_t *h_res);

/* The same two calls for a scan that is still on the device (cc_scan_ingest): no descriptor copy in either direction.
 * cc_db_query_scan answers against the current DB state; cc_db_add_scan = addScan + pushAndBalance. */
int cc_db_query_scan(cc_db *db, cc_scan *scan, const cc_score_t *thres_lb, const cc_score_t *thres_ub, cc_query_result_t *h_res);
int cc_db_add_scan(cc_db *db, cc_scan *scan, double ts, int32_t seed);
/* cc_db_query_scan without the wait, at an explicit epoch (0 .. cc_db_size): *h_res is filled by the next cc_db_query_wait.
 * A per-scan driver that knows its next scans (the evaluator mirror does: test/batch_bin_test.cpp:131-237 walks a list) appends
 * them and queues scan k's query at epoch k while the driver is still busy with scan i < k; the answers are the ones the
 * strictly sequential loop gets (a query at epoch k sees the database as it was after k scans). */
int cc_db_query_scan_submit(cc_db *db, cc_scan *scan, int32_t epoch, const cc_score_t *thres_lb, const cc_score_t *thres_ub,
                            cc_query_result_t *h_res);
/* cc_db_query_wait for the chunks that write into [h_res, h_res + n) only: the caller's own answer, while later submissions
 * stay in flight. */
int cc_db_query_collect(cc_db *db, const cc_query_result_t *h_res, int n);
/* cc_db_add_scans_prepare (the asynchronous first half of an append) for a scan handle; cc_db_add_scan on the same scan
 * later only commits. */
int cc_db_add_scan_prepare(cc_db *db, cc_scan *scan);

/* cc_db_add_scan and cc_db_query_scan_submit for 1..CC_SCAN_BATCH_MAX scan handles at a time: scans[i] is appended with
 * (h_ts[i], h_seed[i]) in the order given; scans[i] is queried at h_epoch[i] (0 .. cc_db_size), its answer goes to h_res[i]
 * (cc_db_query_collect / cc_db_query_wait).  The answers are those of the same calls made one by one; what changes is the
 * number of launches: one chain per batch.  A per-scan driver that knows its next B scans appends them in one call and then
 * queues each one's query at its own position (scan k at epoch k sees the database as it was after k scans): the class
 * mirror's read-ahead does (hostcpp/cont2/contour_db.h). */
int cc_db_add_scan_batch(cc_db *db, cc_scan *const *scans, int n, const double *h_ts, const int32_t *h_seed);
int cc_db_query_scan_batch_submit(cc_db *db, cc_scan *const *scans, int n, const int32_t *h_epoch, const cc_score_t *thres_lb,
                                  const cc_score_t *thres_ub, cc_query_result_t *h_res);

/* The two batched calls with host descriptor buffers (one H2D copy each): for drivers that keep descriptors on the host,
 * e.g. an offline replay of a whole sequence (all scans added, then scan i queried against epoch i). */
int cc_db_add_scans_host(cc_db *db, const cc_scan_desc_t *h_desc, int n, const double *h_ts, const int32_t *h_seed);
int cc_db_query_batch_host(cc_db *db, const cc_scan_desc_t *h_qdesc, int nq, const int32_t *h_epoch,
                           const cc_score_t *thres_lb, const cc_score_t *thres_ub, cc_query_result_t *h_res);

/* CandidateManager driven by explicit anchor hints instead of the KNN search (the single-pair flow of
 * test/kitti_read_bin_test.cpp:226-291): for one query scan, CandidateManager::checkCandWithHint (contour_db.h:374-488)
 * for every hint IN THE GIVEN ORDER, then tidyUpCandidates (:494-596) and fineOptimize (:604-648).  Candidate scans are
 * scans of `db` (any scan added so far, searchable or not).  Hint levels must be within 1..4 (DIST_BIN_LAYERS), at most
 * CC_HINT_MAX hints.  h_scores (optional, [n_hints]) receives what checkCandWithHint returns per hint. */
#define CC_HINT_MAX (CC_NQLEV * CC_NPIV * CC_KNN_MAX)
typedef struct {
  int32_t cand_gidx; /* candidate scan = cm_cand (DB index)                    */
  int8_t level;      /* ConstellationPair{level, seq_src, seq_tgt}, contour_mng.h:221-240 */
  int8_t seq_src;    /* anchor contour of the candidate scan                   */
  int8_t seq_tgt;    /* anchor contour of the query scan                       */
  int8_t pad;
} cc_hint_t;
typedef struct {
  int32_t i_ovlp_sum, i_ovlp_max_one, i_in_ang_rng; /* ScoreConstellSim  */
  int32_t i_indiv_sim, i_orie_sim;                  /* ScorePairwiseSim  */
  int32_t passed;                                   /* 1: a proposal was added for this hint */
} cc_hint_score_t;
int cc_db_check_hints(cc_db *db, const cc_scan_desc_t *d_qdesc, const cc_hint_t *h_hints, int n_hints,
                      const cc_score_t *thres_lb, const cc_score_t *thres_ub, int max_fine_opt,
                      cc_query_result_t *h_res, cc_hint_score_t *h_scores, void *stream);
int cc_db_check_hints_host(cc_db *db, const cc_scan_desc_t *h_qdesc, const cc_hint_t *h_hints, int n_hints,
                           const cc_score_t *thres_lb, const cc_score_t *thres_ub, int max_fine_opt,
                           cc_query_result_t *h_res, cc_hint_score_t *h_scores);

/* Parity / debug: the constellations of the LAST cc_db_check_hints[_host] call that passed all four gates, in hint
 * order: the pose getTFFromConstell returned for each (contour_mng.h:1246-1277, before any proposal merging) and the
 * constellation it was computed from, so that a test can redo the rigid fit independently (e.g. with an SVD). */
typedef struct {
  int32_t hint;          /* index into the call's hint array                                   */
  int32_t n_pairs;       /* contours pairs in the constellation                                */
  double tf[3];          /* T_pass = (x, y, theta), BEV pixel units / radians                  */
  uint64_t pairs[7];     /* the pairs as a set: bit (level-1)*100 + seq_src*10 + seq_tgt       */
} cc_pass_dbg_t;
int cc_db_debug_passes(cc_db *db, cc_pass_dbg_t *h_out, int cap, int *n_out);

/* ---- the compact per-scan records (multi-GPU exchange) ----
 * cc_pack_scans turns full descriptors into the two records the database keeps per scan: the hot record
 * (cc_hot_desc_t, 18 KB) and the correlation inputs (opaque, 41 KB; cc_packed_sizes gives both sizes).  A rank packs
 * the scans it ingested, the ranks all-gather the two arrays over RCCL (59 KB per scan instead of the 169 KB
 * descriptor), and every rank appends the gathered scans to its replica with cc_db_add_packed -- the same effect as
 * cc_db_add_scans on the full descriptors (which is pack + add_packed).  d_hot_out / d_feat_out: device arrays of n
 * records each. */
void cc_packed_sizes(size_t *hot_bytes, size_t *feat_bytes);
int cc_pack_scans(cc_ctx *ctx, const cc_scan_desc_t *d_desc, int n, void *d_hot_out, void *d_feat_out, void *stream);
int cc_db_add_packed(cc_db *db, const void *d_hot, const void *d_feat, int n, const double *h_ts, const int32_t *h_seed,
                     void *stream);
/* Device pointers of the DB's own record arrays ([cc_db_size()] records each). */
const void *cc_db_hot_ptr(const cc_db *db);
const void *cc_db_feat_ptr(const cc_db *db);

/* Same for the query kernels: accumulated ms {K3 knn, K4 check, K4b merge, K5 gmm, K6 final} summed over the chunk
 * launches (chunks in flight together overlap in time), and the number of QUERIES the sums cover (*n_launches).
 * on = 0: off; 1: every chunk carries the six stage events; n > 1: every n-th chunk does (the events cost throughput:
 * ~7 % on the bench when every chunk is timed; the sums are normalised by the queries they cover either way). */
int cc_db_profile_enable(cc_db *db, int on);
int cc_db_profile_read(cc_db *db, double ms_out[5], int *n_launches);

/* cc_db_query_batch cuts a batch into one chunk per lane (<= 1024 queries each; a streamed cc_db_query_submit of 1024
 * queries or more goes out in chunks of 1024, batch after batch on alternating lanes) and keeps up to two chunks in
 * flight on internal streams
 * (the f64-bound correlation of one chunk overlaps the latency-bound retrieval/checks of the next).  n = 1 runs the
 * chunks one after the other (per-kernel timing, debugging); default 2.  No reference counterpart. */
int cc_db_set_lanes(cc_db *db, int n);

/* Host-side introspection of the K0 bookkeeping for parity tests:
 * tree sizes per (layer, bucket) and bucket ranges at the current epoch. */
int cc_db_bucket_state(const cc_db *db, int32_t *tree_sizes /*[3][6]*/, float *ranges /*[3][7]*/);

/* ---- the multi-GPU exchange, owned by the library ----
 * One process per GPU.  The path has ONE collective: the all-gather of the compact per-scan records (cc_pack_scans ->
 * cc_db_add_packed on every rank; 59 KB per scan) over RCCL / xGMI -- ncclAllGather, no all-reduce anywhere.  The reference
 * has no counterpart (it is a single-process CPU library); these calls are what a C++ multi-GPU driver binds
 * (hostcpp/examples/batch_replay_mgpu.cpp) and what bench.py --comm-owner c reaches through ctypes.  RCCL is loaded with
 * dlopen at the first call: a single-GPU process never touches it.
 *   cc_comm_unique_id        : rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the others by any means
 *   cc_comm_create           : ncclCommInitRank on `device` (collective over the world)
 *   cc_comm_create_from_env  : the two above for a launcher that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT
 *                              (torch.distributed.run, batch_replay_mgpu's forker) on ONE node: the id travels through a
 *                              file under /dev/shm named after MASTER_PORT
 *   cc_comm_allgather_packed : d_send = this rank's bytes_per_rank bytes, d_recv = world x bytes_per_rank, rank-major;
 *                              queued on `stream` (hipStream_t)
 * UNMEASURED on hardware with more than one rank: the build boxes have one GPU (world = 1 runs there: tests/test_gpu_comm.py). */
typedef struct cc_comm cc_comm;
int cc_comm_unique_id(void *id128);
int cc_comm_create(int device, int rank, int world, const void *id128, cc_comm **out);
int cc_comm_create_from_env(cc_comm **out, int *rank_out, int *world_out);
int cc_comm_rank(const cc_comm *comm);
int cc_comm_world(const cc_comm *comm);
int cc_comm_allgather_packed(cc_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream);
int cc_comm_destroy(cc_comm *comm);

/* ------------------------------------------------------------ pose helpers (host) ------- */
/* ConstellCorrelation::getEstSensTF (correlation.h:287-296): BEV-frame T_delta -> sensor
 * frame. in/out = (x, y, theta). */
void cc_est_sens_tf(const double tf_bev[3], int n_row, int n_col, double tf_sens[3]);

#ifdef __cplusplus
}
#endif
#endif /* CONT2_AMD_H */
