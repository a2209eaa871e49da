// A C++ multi-GPU driver on the C-ABI alone (BASELINE.json configs 4 and 5): one process per GPU, the
// path's ONE collective owned by the library (cc_comm_allgather_packed = ncclAllGather of the packed per-scan records over
// RCCL / xGMI).  The offline replay of test/batch_bin_test.cpp:131-237 -- every scan is queried against the scans that were
// in the database when it arrived -- sharded the way contour-context_amd/sharding.py shards it for bench.py:
//   * ingest, scan-sharded and interleaved: rank r reads and ingests scans r, r + N, r + 2N, ... (cc_ingest_batch);
//   * each rank packs its scans (cc_pack_scans: 18 KB hot record + 41 KB correlation inputs instead of the 169 KB
//     descriptor) and the ranks all-gather the records; every rank re-orders them to scan order and appends them to its
//     replica (cc_db_add_packed: the host bookkeeping is deterministic, the replicas are identical);
//   * queries, sharded the same way: rank r queries ITS scans, scan i at epoch i (what the database held before scan i was
//     added), against its replica -- no further exchange;
//   * every rank writes `<out_prefix>.rank<r>.txt` (one line per query: scan, matched scan or -1, correlation, x y theta of
//     T_delta in BEV units); the union over the ranks is the replay's outcome.
//
//   g++ -O2 -std=c++17 batch_replay_mgpu.cpp -I../../../include -L../.. -lcont2_amd -Wl,-rpath,$PWD/../.. -L/opt/rocm/lib -lamdhip64 -o batch_replay_mgpu
//   ./batch_replay_mgpu <scan_list.txt> <out_prefix> [--gpus N]
// scan_list.txt: one scan per line, "<timestamp seconds> <path to KITTI .bin>".  --gpus N > 1 without a launcher: the
// program forks N ranks itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT, as torch.distributed.run sets them); under a
// launcher that has set RANK already it is one of the ranks.  UNMEASURED with more than one rank: the build boxes have one
// GPU (N = 1 runs there and goes through RCCL with a world of one: tests/test_gpu_comm.py).
#include <hip/hip_runtime_api.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "cont2_amd.h"

#define CCK(call)                                                                 \
  do {                                                                            \
    const int rc_ = (call);                                                       \
    if (rc_ != CC_OK) {                                                           \
      fprintf(stderr, "rank %d: %s failed (%d): %s\n", g_rank, #call, rc_, cc_last_error()); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)
#define HIPK(call)                                                                 \
  do {                                                                             \
    const hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "rank %d: %s: %s\n", g_rank, #call, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)
static int g_rank = 0;

struct ScanRef {
  double ts;
  std::string path;
};

static int shard_len(int n, int world) { return (n + world - 1) / world; }  // sharding.py:shard_len

static int run_rank(const std::string &list_path, const std::string &out_prefix) {
  cc_comm *comm = nullptr;
  int rank = 0, world = 1;
  const char *el = getenv("LOCAL_RANK");
  const int device = el ? atoi(el) : 0;
  HIPK(hipSetDevice(device));
  CCK(cc_comm_create_from_env(&comm, &rank, &world));
  g_rank = rank;

  std::vector<ScanRef> scans;
  {
    std::ifstream f(list_path);
    std::string line;
    while (std::getline(f, line)) {
      std::istringstream is(line);
      ScanRef s;
      if (is >> s.ts >> s.path) scans.push_back(s);
    }
  }
  const int n = (int)scans.size();
  if (n == 0) {
    fprintf(stderr, "rank %d: empty scan list %s\n", rank, list_path.c_str());
    return 2;
  }
  const int shard = shard_len(n, world);
  std::vector<int> mine;
  for (int i = rank; i < n; i += world) mine.push_back(i);

  cc_manager_cfg_t mcfg;
  cc_db_cfg_t dcfg;
  cc_score_t lb, ub;
  cc_default_manager_cfg(&mcfg);
  cc_default_db_cfg(&dcfg);
  cc_default_thresholds(&lb, &ub);
  const int CH = 64;  // scans per ingest call
  cc_ctx *ctx = nullptr;
  CCK(cc_create(device, &mcfg, CH, &ctx));
  size_t HB = 0, FB = 0;
  cc_packed_sizes(&HB, &FB);
  const size_t REC = HB + FB;

  // ---- scan-sharded ingest + pack
  cc_scan_desc_t *d_desc = nullptr;  // this rank's descriptors (kept: they are its queries)
  char *d_hot = nullptr, *d_feat = nullptr, *d_rec_local = nullptr, *d_rec_all = nullptr;
  HIPK(hipMalloc(&d_desc, sizeof(cc_scan_desc_t) * (size_t)(mine.empty() ? 1 : mine.size())));
  HIPK(hipMalloc(&d_hot, HB * (size_t)shard));
  HIPK(hipMalloc(&d_feat, FB * (size_t)shard));
  HIPK(hipMalloc(&d_rec_local, REC * (size_t)shard));
  HIPK(hipMalloc(&d_rec_all, REC * (size_t)shard * world));
  HIPK(hipMemset(d_rec_local, 0, REC * (size_t)shard));  // padding rows of a short last shard
  const size_t cap_pts = 1000000 / 4;  // readKITTIPointCloudBin reads at most 1 000 000 floats (tools/pointcloud_util.h:9-47)
  std::vector<float> h_pts(4 * cap_pts * CH);
  float *d_pts = nullptr;
  HIPK(hipMalloc(&d_pts, sizeof(float) * 4 * cap_pts * CH));
  for (size_t c0 = 0; c0 < mine.size(); c0 += CH) {
    const int nb = (int)std::min<size_t>(CH, mine.size() - c0);
    std::vector<int64_t> offs(nb + 1, 0);
    for (int k = 0; k < nb; k++) {
      FILE *f = fopen(scans[mine[c0 + k]].path.c_str(), "rb");
      if (!f) {
        printf("Lidar bin file %s does not exist.\n", scans[mine[c0 + k]].path.c_str());
        exit(-1);
      }
      const size_t np = fread(h_pts.data() + 4 * offs[k], 4 * sizeof(float), cap_pts, f);
      fclose(f);
      offs[k + 1] = offs[k] + (int64_t)np;
    }
    HIPK(hipMemcpy(d_pts, h_pts.data(), sizeof(float) * 4 * (size_t)offs[nb], hipMemcpyHostToDevice));
    CCK(cc_ingest_batch(ctx, d_pts, offs.data(), nb, d_desc + c0, nullptr, nullptr));
    CCK(cc_pack_scans(ctx, d_desc + c0, nb, d_hot + HB * c0, d_feat + FB * c0, nullptr));
  }
  // one record per scan: hot | feat
  if (!mine.empty()) {
    HIPK(hipMemcpy2D(d_rec_local, REC, d_hot, HB, HB, mine.size(), hipMemcpyDeviceToDevice));
    HIPK(hipMemcpy2D(d_rec_local + HB, REC, d_feat, FB, FB, mine.size(), hipMemcpyDeviceToDevice));
  }
  HIPK(hipDeviceSynchronize());

  // ---- the exchange: ONE all-gather of the packed records
  hipEvent_t e0, e1;
  HIPK(hipEventCreate(&e0));
  HIPK(hipEventCreate(&e1));
  HIPK(hipEventRecord(e0, nullptr));
  CCK(cc_comm_allgather_packed(comm, d_rec_local, d_rec_all, REC * (size_t)shard, nullptr));
  HIPK(hipEventRecord(e1, nullptr));
  HIPK(hipEventSynchronize(e1));
  float ms_x = 0.f;
  HIPK(hipEventElapsedTime(&ms_x, e0, e1));

  // ---- scan order: scan i = r + world * j sits in row r * shard + j of the gathered array (sharding.py:scan_order)
  char *d_hot_all = nullptr, *d_feat_all = nullptr;
  HIPK(hipMalloc(&d_hot_all, HB * (size_t)n));
  HIPK(hipMalloc(&d_feat_all, FB * (size_t)n));
  for (int r = 0; r < world; r++) {
    const int cnt = (n - r + world - 1) / world;  // scans of rank r
    if (cnt <= 0) continue;
    const char *src = d_rec_all + REC * (size_t)shard * r;
    HIPK(hipMemcpy2D(d_hot_all + HB * (size_t)r, HB * (size_t)world, src, REC, HB, cnt, hipMemcpyDeviceToDevice));
    HIPK(hipMemcpy2D(d_feat_all + FB * (size_t)r, FB * (size_t)world, src + HB, REC, FB, cnt, hipMemcpyDeviceToDevice));
  }
  cc_db *db = nullptr;
  CCK(cc_db_create(ctx, &dcfg, n + 16, &db));
  std::vector<double> ts(n);
  std::vector<int32_t> ids(n);
  for (int i = 0; i < n; i++) {
    ts[i] = scans[i].ts;
    ids[i] = i;
  }
  CCK(cc_db_add_packed(db, d_hot_all, d_feat_all, n, ts.data(), ids.data(), nullptr));

  // ---- query-sharded scoring: this rank's scans, scan i at epoch i
  std::vector<int32_t> epochs(mine.size());
  for (size_t k = 0; k < mine.size(); k++) epochs[k] = mine[k];
  std::vector<cc_query_result_t> res(mine.size());
  if (!mine.empty()) CCK(cc_db_query_batch(db, d_desc, (int)mine.size(), epochs.data(), &lb, &ub, res.data(), nullptr, nullptr, nullptr));
  int n_loop = 0;
  {
    const std::string op = out_prefix + ".rank" + std::to_string(rank) + ".txt";
    FILE *f = fopen(op.c_str(), "w");
    if (!f) {
      fprintf(stderr, "rank %d: cannot write %s\n", rank, op.c_str());
      return 2;
    }
    for (size_t k = 0; k < mine.size(); k++) {
      const cc_query_result_t &q = res[k];
      n_loop += q.n_res > 0;
      fprintf(f, "%d %d %.6f %.6f %.6f %.6f\n", mine[k], q.n_res > 0 ? q.cand_gidx : -1, q.correlation, q.tf[0], q.tf[1], q.tf[2]);
    }
    fclose(f);
  }
  const size_t gathered = REC * (size_t)shard * world;
  printf("rank %d of %d: %zu scans ingested and queried, %d loop closures; exchange %zu bytes gathered in %.3f ms (%.1f GB/s received from peers)\n",
         rank, world, mine.size(), n_loop, gathered, ms_x, world > 1 ? gathered * (double)(world - 1) / world / (ms_x * 1e-3) / 1e9 : 0.0);
  cc_db_destroy(db);
  cc_comm_destroy(comm);
  cc_destroy(ctx);
  (void)hipFree(d_desc);
  (void)hipFree(d_hot);
  (void)hipFree(d_feat);
  (void)hipFree(d_rec_local);
  (void)hipFree(d_rec_all);
  (void)hipFree(d_hot_all);
  (void)hipFree(d_feat_all);
  (void)hipFree(d_pts);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <scan_list.txt> <out_prefix> [--gpus N]\n", argv[0]);
    return 2;
  }
  int gpus = 1;
  for (int i = 3; i + 1 < argc; i++)
    if (!strcmp(argv[i], "--gpus")) gpus = atoi(argv[i + 1]);
  if (getenv("RANK") || gpus <= 1) return run_rank(argv[1], argv[2]);  // under a launcher, or a world of one
  // our own launcher: one process per GPU, forked BEFORE anything touches HIP
  char port[32];
  snprintf(port, sizeof(port), "%d", 20000 + (int)(getpid() % 20000));
  char token[64];
  snprintf(token, sizeof(token), "%d-%ld", (int)getpid(), (long)time(nullptr));
  std::vector<pid_t> kids;
  for (int r = 0; r < gpus; r++) {
    const pid_t p = fork();
    if (p == 0) {
      char b[32];
      snprintf(b, sizeof(b), "%d", r);
      setenv("RANK", b, 1);
      setenv("LOCAL_RANK", b, 1);
      snprintf(b, sizeof(b), "%d", gpus);
      setenv("WORLD_SIZE", b, 1);
      setenv("MASTER_PORT", port, 1);
      setenv("CC_COMM_TOKEN", token, 1);  // names this job's id file (cc_comm_create_from_env)
      setenv("MASTER_ADDR", "127.0.0.1", 1);
      setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);  // dmabuf IPC between the ranks of one node
      return run_rank(argv[1], argv[2]);
    }
    kids.push_back(p);
  }
  int bad = 0;
  for (pid_t p : kids) {
    int st = 0;
    waitpid(p, &st, 0);
    bad |= !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
  }
  return bad;
}
