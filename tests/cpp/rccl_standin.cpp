// TEST INFRASTRUCTURE ONLY -- never linked into or loaded by the product.
//
// A stand-in TRANSPORT for the fourteen RCCL entry points libeigenex_hip.so calls, so that the library's multi-rank code
// (request-list exchange, packed neighbour exchange inside ncclGroupStart/End, all-reduce schedule, phase all-gather, the
// second communicator of the overlapped exchange) can run BETWEEN DIFFERENT PROCESSES on a box with one GPU, where RCCL
// itself refuses two ranks on one device ("Duplicate GPU detected").  The tests preload it (LD_PRELOAD) into rank
// processes that all use device 0.  Nothing here is RCCL and nothing measured with it is a performance number: data goes
// device -> host -> a file in a shared directory -> host -> device, synchronously.
//
// Semantics kept from NCCL, because the library relies on them:
//   * operations are ordered with the stream they are given (here: the stream is drained first, the copy back is
//     synchronous, so later work on the stream sees the data);
//   * send/recv inside a group are deferred to ncclGroupEnd, where all sends are posted before any receive is awaited
//     (two ranks that both "send then receive" do not deadlock);
//   * every rank must issue the collectives of one communicator in the same order (sequence numbers per communicator;
//     a rank that does not fails by TIME-OUT, ncclSystemError, instead of hanging the test);
//   * ncclSum over doubles adds the contributions in rank order 0,1,..,n-1 on every rank: all ranks get the same bits,
//     and for two ranks the same bits as any other order.
//
// Directory: $EIGENEX_TEST_RCCL_DIR (required).  Files are never reused (sequence numbers) and are left for the test to
// delete with its temporary directory.
//
// The HIP runtime is NOT linked: a preloaded library that pulled in the system libamdhip64 ahead of the copy a PyTorch
// wheel bundles would put two runtimes into the process (cmpt_eigenex_amd.capi refuses that).  The three HIP calls used here
// are looked up at first use in whichever libamdhip64.so.7 the process has loaded by then.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Comm {
  int n = 1, rank = 0, device = 0;
  std::string dir;
  long coll_seq = 0;
  int splits = 0;
  std::map<int, long> sent, received;  // per peer
};

struct P2P {
  bool send;
  void* buf;
  size_t bytes;
  int peer;
  Comm* comm;
  hipStream_t stream;
};

struct Hip {
  hipError_t (*stream_synchronize)(hipStream_t) = nullptr;
  hipError_t (*memcpy)(void*, const void*, size_t, hipMemcpyKind) = nullptr;
  hipError_t (*get_device)(int*) = nullptr;
  Hip() {
    void* h = dlopen("libamdhip64.so.7", RTLD_NOLOAD | RTLD_NOW);
    if (!h) h = dlopen("libamdhip64.so", RTLD_NOLOAD | RTLD_NOW);
    if (!h) {
      std::fprintf(stderr, "rccl_standin: no HIP runtime is loaded in this process\n");
      std::abort();
    }
    stream_synchronize = reinterpret_cast<decltype(stream_synchronize)>(dlsym(h, "hipStreamSynchronize"));
    memcpy = reinterpret_cast<decltype(memcpy)>(dlsym(h, "hipMemcpy"));
    get_device = reinterpret_cast<decltype(get_device)>(dlsym(h, "hipGetDevice"));
    if (!stream_synchronize || !memcpy || !get_device) std::abort();
  }
};
const Hip& hip() {
  static const Hip h;
  return h;
}

thread_local int g_depth = 0;
thread_local std::vector<P2P> g_queue;

size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

double timeout_seconds() {
  const char* e = std::getenv("EIGENEX_TEST_RCCL_TIMEOUT");
  return e ? std::atof(e) : 120.0;
}

bool write_file(const std::string& path, const void* data, size_t bytes) {
  const std::string tmp = path + ".part";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = bytes == 0 || std::fwrite(data, 1, bytes, f) == bytes;
  std::fclose(f);
  return ok && std::rename(tmp.c_str(), path.c_str()) == 0;  // rename: readers never see a partial file
}

bool read_file(const std::string& path, void* data, size_t bytes) {
  const auto t0 = std::chrono::steady_clock::now();
  struct stat st;
  while (stat(path.c_str(), &st) != 0) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_seconds()) {
      std::fprintf(stderr, "rccl_standin: timed out waiting for %s\n", path.c_str());
      return false;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  if ((size_t)st.st_size != bytes) {
    std::fprintf(stderr, "rccl_standin: %s holds %zu bytes, the receiver expects %zu\n", path.c_str(), (size_t)st.st_size, bytes);
    return false;
  }
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  const bool ok = bytes == 0 || std::fread(data, 1, bytes, f) == bytes;
  std::fclose(f);
  return ok;
}

ncclResult_t to_host(std::vector<char>& h, const void* dev, size_t bytes, hipStream_t s) {
  if (hip().stream_synchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  h.resize(bytes);
  if (bytes && hip().memcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

ncclResult_t to_device(void* dev, const std::vector<char>& h, size_t bytes) {
  if (bytes && hip().memcpy(dev, h.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

ncclResult_t do_send(const P2P& p) {
  std::vector<char> h;
  if (ncclResult_t r = to_host(h, p.buf, p.bytes, p.stream)) return r;
  const long seq = p.comm->sent[p.peer]++;
  const std::string path = p.comm->dir + "/p2p_" + std::to_string(p.comm->rank) + "_" + std::to_string(p.peer) + "_" + std::to_string(seq);
  return write_file(path, h.data(), p.bytes) ? ncclSuccess : ncclSystemError;
}

ncclResult_t do_recv(const P2P& p) {
  if (hip().stream_synchronize(p.stream) != hipSuccess) return ncclUnhandledCudaError;
  const long seq = p.comm->received[p.peer]++;
  const std::string path = p.comm->dir + "/p2p_" + std::to_string(p.peer) + "_" + std::to_string(p.comm->rank) + "_" + std::to_string(seq);
  std::vector<char> h(p.bytes);
  if (!read_file(path, h.data(), p.bytes)) return ncclSystemError;
  return to_device(p.buf, h, p.bytes);
}

ncclResult_t p2p(bool send, void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || peer < 0 || peer >= c->n || !type_size(t)) return ncclInvalidArgument;
  const P2P p{send, buf, count * type_size(t), peer, c, s};
  if (g_depth > 0) {
    g_queue.push_back(p);
    return ncclSuccess;
  }
  return send ? do_send(p) : do_recv(p);
}

// every rank's contribution to collective number `seq`, in rank order
ncclResult_t exchange_all(Comm* c, const char* tag, const void* send, size_t bytes, hipStream_t s, std::vector<std::vector<char>>& all) {
  std::vector<char> mine;
  if (ncclResult_t r = to_host(mine, send, bytes, s)) return r;
  const long seq = c->coll_seq++;
  const std::string stem = c->dir + "/" + tag + "_" + std::to_string(seq) + "_";
  if (!write_file(stem + std::to_string(c->rank), mine.data(), bytes)) return ncclSystemError;
  all.assign(c->n, std::vector<char>(bytes));
  for (int r = 0; r < c->n; ++r) {
    if (r == c->rank) all[r] = mine;
    else if (!read_file(stem + std::to_string(r), all[r].data(), bytes)) return ncclSystemError;
  }
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof(*id));
  const long long stamp = std::chrono::steady_clock::now().time_since_epoch().count();
  std::snprintf(id->internal, sizeof(id->internal), "standin-%ld-%llx", (long)getpid(), stamp);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  const char* base = std::getenv("EIGENEX_TEST_RCCL_DIR");
  if (!base || !comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm* c = new Comm;
  c->n = nranks;
  c->rank = rank;
  (void)hip().get_device(&c->device);
  id.internal[sizeof(id.internal) - 1] = 0;
  c->dir = std::string(base) + "/" + id.internal;
  mkdir(c->dir.c_str(), 0700);  // EEXIST from the other ranks is fine
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t* newcomm, ncclConfig_t*) {
  Comm* p = reinterpret_cast<Comm*>(comm);
  if (!p || !newcomm || color != 0 || key != p->rank) return ncclInvalidArgument;  // the one use the library makes of it
  Comm* c = new Comm;
  c->n = p->n;
  c->rank = p->rank;
  c->device = p->device;
  c->dir = p->dir + "/split" + std::to_string(p->splits++);
  mkdir(c->dir.c_str(), 0700);
  *newcomm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete reinterpret_cast<Comm*>(comm);
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
  *count = reinterpret_cast<const Comm*>(comm)->n;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
  *rank = reinterpret_cast<const Comm*>(comm)->rank;
  return ncclSuccess;
}

ncclResult_t ncclCommCuDevice(const ncclComm_t comm, int* device) {
  *device = reinterpret_cast<const Comm*>(comm)->device;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "rccl_standin: a peer's message did not arrive (time-out) or the shared directory failed";
    case ncclInvalidArgument: return "rccl_standin: invalid argument";
    case ncclUnhandledCudaError: return "rccl_standin: HIP error";
    default: return "rccl_standin: error";
  }
}

ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  std::vector<P2P> q;
  q.swap(g_queue);
  ncclResult_t res = ncclSuccess;
  for (const P2P& p : q)
    if (p.send && res == ncclSuccess) res = do_send(p);
  for (const P2P& p : q)
    if (!p.send && res == ncclSuccess) res = do_recv(p);
  return res;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  return p2p(true, const_cast<void*>(buf), count, t, peer, comm, s);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  return p2p(false, buf, count, t, peer, comm, s);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || op != ncclSum || (t != ncclFloat64 && t != ncclInt32)) return ncclInvalidArgument;
  const size_t bytes = count * type_size(t);
  std::vector<std::vector<char>> all;
  if (ncclResult_t r = exchange_all(c, "ar", send, bytes, s, all)) return r;
  std::vector<char> out = all[0];
  for (int r = 1; r < c->n; ++r)
    for (size_t i = 0; i < count; ++i) {
      if (t == ncclFloat64) reinterpret_cast<double*>(out.data())[i] += reinterpret_cast<const double*>(all[r].data())[i];
      else reinterpret_cast<int32_t*>(out.data())[i] += reinterpret_cast<const int32_t*>(all[r].data())[i];
    }
  return to_device(recv, out, bytes);
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || !type_size(t)) return ncclInvalidArgument;
  const size_t bytes = sendcount * type_size(t);
  std::vector<std::vector<char>> all;
  if (ncclResult_t r = exchange_all(c, "ag", send, bytes, s, all)) return r;
  for (int r = 0; r < c->n; ++r)
    if (ncclResult_t e = to_device(static_cast<char*>(recv) + (size_t)r * bytes, all[r], bytes)) return e;
  return ncclSuccess;
}

// so that a test can tell which transport a process ended up with
int eigenex_test_rccl_standin_loaded() { return 1; }

}  // extern "C"
