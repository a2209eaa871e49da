// TEST INFRASTRUCTURE: the five RCCL entry points csrc/cc_comm.inc binds (ncclGetUniqueId, ncclCommInitRank, ncclAllGather,
// ncclCommDestroy, ncclGetErrorString) over POSIX shared memory, so that the C++-owned collective of the path -- the id file
// of cc_comm_create_from_env, the rank order of cc_comm_allgather_packed, batch_replay_mgpu's own forker, a last shard that
// is shorter than the others -- runs with MORE THAN ONE RANK on a machine without GPUs (tests/test_cpp_collective_ranks.py;
// the CPU harness' "device" memory is host memory).  Loaded through CC_RCCL_LIB by the harness build only; nothing of it
// is in the product, and it says nothing about RCCL's performance or xGMI.
//   g++ -O1 -std=c++17 -fPIC -shared nccl_standin.cpp -lrt -o libnccl_standin.so
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

namespace {
struct Shared {                 // one segment per communicator, created by whoever comes first
  std::atomic<int> init;        // 0 -> 1 (being set up) -> 2 (ready)
  std::atomic<int> arrived[2];  // sense-reversing barrier: counter of the current / next phase
  std::atomic<int> phase;
  std::atomic<long long> seq;   // collectives completed (statistics)
};
struct Comm {
  Shared *sh = nullptr;
  int rank = 0, world = 1, my_phase = 0;
  char name[96];
  long long n_coll = 0;
};
struct Uid {
  char b[128];
};

void nap() { usleep(200); }

// every rank calls it the same number of times; returns when all `world` ranks have
void barrier(Comm *c) {
  Shared *s = c->sh;
  const int p = c->my_phase & 1;
  if (s->arrived[p].fetch_add(1) + 1 == c->world) {
    s->arrived[p].store(0);
    s->phase.fetch_add(1);
  } else {
    const int want = c->my_phase + 1;
    while (s->phase.load() < want) nap();
  }
  c->my_phase++;
}

void *map_segment(const char *name, size_t bytes, bool *created) {
  int fd = shm_open(name, O_RDWR | O_CREAT | O_EXCL, 0600);
  *created = fd >= 0;
  if (fd < 0) {
    for (int t = 0; t < 50000 && fd < 0; t++) {  // 10 s
      fd = shm_open(name, O_RDWR, 0600);
      if (fd < 0) nap();
    }
    if (fd < 0) return nullptr;
    struct stat st;
    for (int t = 0; t < 50000; t++) {  // the creator's ftruncate
      if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break;
      nap();
    }
  } else if (ftruncate(fd, (off_t)bytes) != 0) {
    close(fd);
    return nullptr;
  }
  void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  return p == MAP_FAILED ? nullptr : p;
}
}  // namespace

extern "C" {
int ncclGetUniqueId(Uid *id) {
  memset(id, 0, sizeof(*id));
  unsigned long long r = 0;
  FILE *f = fopen("/dev/urandom", "rb");
  if (f) {
    if (fread(&r, sizeof(r), 1, f) != 1) r = 0;
    fclose(f);
  }
  snprintf(id->b, sizeof(id->b), "/cc_nccl_standin_%d_%llx_%lx", (int)getpid(), r, (long)time(nullptr));
  return 0;
}

int ncclCommInitRank(Comm **out, int world, Uid id, int rank) {
  if (!out || world < 1 || rank < 0 || rank >= world || id.b[0] != '/') return 4;  // ncclInvalidArgument
  Comm *c = new Comm();
  c->rank = rank;
  c->world = world;
  snprintf(c->name, sizeof(c->name), "%.90s", id.b);
  bool created = false;
  c->sh = (Shared *)map_segment(c->name, sizeof(Shared), &created);
  if (!c->sh) {
    delete c;
    return 2;  // ncclSystemError
  }
  if (created) {
    c->sh->arrived[0].store(0);
    c->sh->arrived[1].store(0);
    c->sh->phase.store(0);
    c->sh->seq.store(0);
    c->sh->init.store(2);
  } else {
    while (c->sh->init.load() != 2) nap();
  }
  barrier(c);  // ncclCommInitRank is a collective
  *out = c;
  return 0;
}

// count elements of `dtype` per rank (1 = ncclUint8: bytes); recv is rank-major; the stream is the harness' (synchronous)
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, Comm *c, void * /*stream*/) {
  if (!c || !send || !recv || dtype != 1) return 4;
  char nm[128];
  snprintf(nm, sizeof(nm), "%.90s_c%lld", c->name, c->n_coll);
  bool created = false;
  char *seg = (char *)map_segment(nm, count * (size_t)c->world, &created);
  if (!seg) return 2;
  memcpy(seg + count * (size_t)c->rank, send, count);
  barrier(c);  // every rank's part is in the segment
  memcpy(recv, seg, count * (size_t)c->world);
  barrier(c);  // everybody has read it
  munmap(seg, count * (size_t)c->world);
  if (c->rank == 0) {
    shm_unlink(nm);
    c->sh->seq.fetch_add(1);
  }
  c->n_coll++;
  return 0;
}

int ncclCommDestroy(Comm *c) {
  if (!c) return 0;
  barrier(c);
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->sh, sizeof(Shared));
  delete c;
  return 0;
}

const char *ncclGetErrorString(int rc) { return rc == 0 ? "no error" : rc == 2 ? "stand-in: shared-memory segment unavailable" : "stand-in: invalid argument"; }
}
