// The sharded search step behind the C ABI (rf_comm_init / rf_search_sharded; 8e: row shards,
// one all-gather of {scores, ids, flags}, merge on every rank).  The reference is single-process
// (vector_rag_mcp/main.py:51-57 asks one Milvus server); a host that shards the corpus over N GPUs -- one
// process per GPU -- calls, per rank:
//
//   rf_comm_unique_id(id)                 rank 0 only; the host ships the 128 bytes to the other ranks
//   rf_comm_init(rank, world, id, device, &comm)
//   rf_search_sharded(index, comm, q, B, k, ...)     every rank, same B and k, same order of calls
//   rf_comm_destroy(comm)
//
// rf_search_sharded is four enqueues on the caller's stream and no host synchronisation: rf_search into the
// packed send buffer, rf_map_ids (when the shard has an id table), ncclAllGather, rf_merge_shards_packed.
// RCCL is bound at run time (dlopen): the library loads on hosts that have no RCCL, and a process that already
// has one loaded (PyTorch ships its own librccl.so) uses THAT copy -- two RCCL instances in one process do not
// share their device state.
#include "rf_internal.h"
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {
struct NcclId {
  char internal[128];
};
typedef int (*fn_get_unique_id)(NcclId*);
typedef int (*fn_comm_init_rank)(void**, int, NcclId, int);   // the id is passed BY VALUE
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_comm_destroy)(void*);
typedef const char* (*fn_error_string)(int);
constexpr int kNcclInt64 = 4;   // ncclDataType_t: int8 0, uint8 1, int32 2, uint32 3, int64 4

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_error_string error_string = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  const char* env = getenv("RAGFIN_RCCL_PATH");
  if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
  // a copy the process has loaded already (by any path that ends in the soname) wins over a second one
  for (int i = 0; i < 2 && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
  for (int i = 0; i < 3 && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) return;
  Rccl r;
  r.handle = h;
  r.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
  r.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
  r.all_gather = (fn_all_gather)dlsym(h, "ncclAllGather");
  r.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  r.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
  if (r.get_unique_id && r.comm_init_rank && r.all_gather && r.comm_destroy) g_rccl = r;
}

const Rccl* rccl() {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.handle) {
    rf_set_error("RCCL is not available: librccl.so could not be loaded (set RAGFIN_RCCL_PATH)");
    return nullptr;
  }
  return &g_rccl;
}

int nccl_failed(const Rccl* r, const char* what, int rc) {
  rf_set_error("%s failed: %s (%d)", what, r->error_string ? r->error_string(rc) : "?", rc);
  return RF_ERR_HIP;
}
}  // namespace

struct rf_comm {
  void* comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

extern "C" int rf_comm_unique_id(void* id_out) {
  if (!id_out) {
    rf_set_error("rf_comm_unique_id: null output");
    return RF_ERR_INVALID;
  }
  const Rccl* r = rccl();
  if (!r) return RF_ERR_UNSUPPORTED;
  NcclId id;
  const int rc = r->get_unique_id(&id);
  if (rc) return nccl_failed(r, "ncclGetUniqueId", rc);
  memcpy(id_out, id.internal, sizeof id.internal);
  return RF_OK;
}

extern "C" int rf_comm_init(int rank, int world, const void* id, int device, rf_comm_t** out) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) {
    rf_set_error("rf_comm_init: bad argument (rank %d of %d)", rank, world);
    return RF_ERR_INVALID;
  }
  *out = nullptr;
  const Rccl* r = rccl();
  if (!r) return RF_ERR_UNSUPPORTED;
  int prev = 0;
  RF_HIP(hipGetDevice(&prev));
  RF_HIP(hipSetDevice(device));
  NcclId nid;
  memcpy(nid.internal, id, sizeof nid.internal);
  rf_comm* c = new rf_comm();
  c->rank = rank;
  c->world = world;
  c->device = device;
  const int rc = r->comm_init_rank(&c->comm, world, nid, rank);
  (void)hipSetDevice(prev);
  if (rc) {
    delete c;
    return nccl_failed(r, "ncclCommInitRank", rc);
  }
  *out = c;
  return RF_OK;
}

extern "C" int rf_comm_destroy(rf_comm_t* c) {
  if (!c) return RF_OK;
  if (c->comm && g_rccl.comm_destroy) g_rccl.comm_destroy(c->comm);
  delete c;
  return RF_OK;
}

extern "C" int rf_comm_rank(const rf_comm_t* c) { return c ? c->rank : -1; }
extern "C" int rf_comm_world(const rf_comm_t* c) { return c ? c->world : 0; }

// scratch layout, int64 words: [send buffer: words(B, k)][gathered: world x words(B, k)][local fp32 scores: (B k + 1) / 2]
extern "C" size_t rf_search_sharded_scratch_words(const rf_comm_t* c, int B, int k) {
  if (!c || B <= 0 || k <= 0) return 0;
  return ((size_t)c->world + 1) * rf_packed_shard_words(B, k) + ((size_t)B * k + 1) / 2;
}

extern "C" int rf_search_sharded(const rf_index_t* ix, rf_comm_t* c, const void* q_dev, int B, int k, int64_t id_base,
                                 const int64_t* id_map_dev, int64_t n_map, float* scores_dev, int64_t* ids_dev,
                                 uint32_t* flags_dev, void* workspace_dev, size_t workspace_bytes, int64_t* scratch_dev,
                                 size_t scratch_words, void* stream) {
  if (!ix || !c || !q_dev || !scores_dev || !ids_dev || !scratch_dev || B <= 0 || k <= 0) {
    rf_set_error("rf_search_sharded: bad argument");
    return RF_ERR_INVALID;
  }
  if (scratch_words < rf_search_sharded_scratch_words(c, B, k)) {
    rf_set_error("rf_search_sharded: scratch too small (%zu < %zu words)", scratch_words, rf_search_sharded_scratch_words(c, B, k));
    return RF_ERR_CAPACITY;
  }
  if ((id_map_dev == nullptr) != (n_map == 0) || n_map < 0) {
    rf_set_error("rf_search_sharded: id_map_dev and n_map must be given together");
    return RF_ERR_INVALID;
  }
  const Rccl* r = rccl();
  if (!r) return RF_ERR_UNSUPPORTED;
  const size_t words = rf_packed_shard_words(B, k);
  const size_t bk = (size_t)B * k;
  int64_t* const send = scratch_dev;
  int64_t* const gathered = scratch_dev + words;
  float* const local_scores = (float*)(scratch_dev + ((size_t)c->world + 1) * words);
  // 1. the local shard's exact top-k straight into the send buffer {fp64 scores, ids, flags}
  int rc = rf_search(ix, q_dev, B, k, id_map_dev ? 0 : id_base, local_scores, send + bk, (double*)send, (uint32_t*)(send + 2 * bk),
                     workspace_dev, workspace_bytes, stream);
  if (rc) return rc;
  // 2. local row numbers -> global ids where the shard is not a contiguous range
  if (id_map_dev) {
    rc = rf_map_ids(send + bk, (int64_t)bk, id_map_dev, n_map, stream);
    if (rc) return rc;
  }
  // 3. the step's ONE collective (a single rank gathers from itself: the same call, so that the path a
  //    one-GPU test exercises is the path N ranks run)
  rc = r->all_gather(send, gathered, words, kNcclInt64, c->comm, (hipStream_t)stream);
  if (rc) return nccl_failed(r, "ncclAllGather", rc);
  // 4. merge by (score desc, id asc), flags OR-ed over the shards: identical on every rank
  return rf_merge_shards_packed(gathered, c->world, B, k, scores_dev, ids_dev, flags_dev, stream);
}
