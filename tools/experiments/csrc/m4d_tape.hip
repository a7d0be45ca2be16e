// Launch tape: a recorded sequence of kernel launches of this library, replayed as plain stream launches.
//
// Why not hipGraph: a kernel issued by a hipGraph replay -- on either side -- makes small concurrent launches crawl: a chain
// of 30 small-map convolutions (10 us each) takes 340 us beside a queue of chip-filling Winograd layers when BOTH are
// plain stream launches, and ~700 us as soon as either of the two is a graph replay, whatever the stream priority or the
// number of hardware queues (tools/stream_vs_graph_probe.py, profiles/r03_stream_vs_graph_probe.txt).  The frame pipeline
// lives on exactly that concurrency (coarse-level chains of one frame under the level-1 convolutions of the previous one).
// A tape gives the host-side cost of a graph replay (one C loop, ~3 us per launch, no Python) with stream-launch semantics.
//
// m4d_launch() (m4d_common.h) diverts every launch of the recording thread here.  Arguments are copied by value at record
// time (pointers stay pointers: the recording pass must run on the buffers the replays will use -- network.TapedSequence
// records under a torch graph capture, whose private memory pool keeps every intermediate tensor's address).
#include <cstring>
#include <mutex>
#include <vector>
#include "m4d_common.h"
#include "../../../include/m4depth_hip.h"
#include "../../../include/m4depth_hip_experiments.h"

namespace {

struct TapeOp {
  const void* fn;
  dim3 grid, block;
  unsigned lds;
  std::vector<unsigned char> blob;      // argument values, each at its natural alignment
  std::vector<unsigned> offs;           // offset of every argument in the blob
};
struct Tape { std::vector<TapeOp> ops; };

std::mutex g_mu;
std::vector<Tape*> g_tapes;
thread_local Tape* t_rec = nullptr;

}  // namespace

bool m4d_tape_recording() { return t_rec != nullptr; }

void m4d_tape_push(const void* fn, dim3 grid, dim3 block, unsigned lds, void* const* params, const size_t* sizes, int n) {
  TapeOp op;
  op.fn = fn; op.grid = grid; op.block = block; op.lds = lds;
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    const size_t al = sizes[i] >= 16 ? 16 : (sizes[i] >= 8 ? 8 : (sizes[i] >= 4 ? 4 : (sizes[i] >= 2 ? 2 : 1)));
    off = (off + al - 1) / al * al;
    op.offs.push_back((unsigned)off);
    off += sizes[i];
  }
  op.blob.resize(off + 16);
  for (int i = 0; i < n; ++i) std::memcpy(op.blob.data() + op.offs[i], params[i], sizes[i]);
  t_rec->ops.push_back(std::move(op));
}

extern "C" int m4d_tape_begin() {
  if (t_rec != nullptr) return -1;
  Tape* t = new Tape();
  std::lock_guard<std::mutex> lk(g_mu);
  g_tapes.push_back(t);
  t_rec = t;
  return (int)g_tapes.size() - 1;
}

extern "C" int m4d_tape_end() {
  if (t_rec == nullptr) return -1;
  const int n = (int)t_rec->ops.size();
  t_rec = nullptr;
  return n;
}

extern "C" int m4d_tape_length(int tape) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (tape < 0 || tape >= (int)g_tapes.size() || g_tapes[tape] == nullptr) return -1;
  return (int)g_tapes[tape]->ops.size();
}

extern "C" int m4d_tape_replay(int tape, void* stream) {
  Tape* t;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (tape < 0 || tape >= (int)g_tapes.size() || g_tapes[tape] == nullptr) return (int)hipErrorInvalidValue;
    t = g_tapes[tape];
  }
  if (t == t_rec) return (int)hipErrorInvalidValue;                  // still recording
  std::vector<void*> params;                                         // sized per launch: no argument-count limit
  for (TapeOp& op : t->ops) {
    const size_t n = op.offs.size();
    params.resize(n + 1);
    for (size_t i = 0; i < n; ++i) params[i] = op.blob.data() + op.offs[i];
    const hipError_t e = hipLaunchKernel(op.fn, op.grid, op.block, params.data(), op.lds, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

extern "C" int m4d_tape_free(int tape) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (tape < 0 || tape >= (int)g_tapes.size() || g_tapes[tape] == nullptr || g_tapes[tape] == t_rec) return (int)hipErrorInvalidValue;
  delete g_tapes[tape];
  g_tapes[tape] = nullptr;
  return 0;
}
