// Host side of the streamed ingest and of the streamed results (config C5): batched positional reads into / writes
// from the caller's (page-locked) blocks.
// The reference reads one TIFF page per dask block through tifffile (reader.py:265-292); here the pages of a chunk
// of timepoints are a list of byte runs that a handful of threads read side by side, without the interpreter
// (the Python reader spent two thirds of its time handing 512 pages per chunk through futures and the GIL).
#include <errno.h>
#include <unistd.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/magnify_hip.h"

namespace {

struct run_list {
  const int32_t* fds;
  const int64_t* offsets;
  const int64_t* nbytes;
  void* const* dsts;
  int n;
  std::atomic<int> next{0};
  std::atomic<int> failed_run{-1};
  std::atomic<int> failed_errno{0};
};

// a run is read in pieces: a piece that large keeps a thread busy for ~1 ms, so the last pieces of a chunk spread
constexpr int64_t PIECE = 8 << 20;

template <bool WRITE>
void run_worker(run_list* rl) {
  for (;;) {
    const int i = rl->next.fetch_add(1, std::memory_order_relaxed);
    if (i >= rl->n || rl->failed_run.load(std::memory_order_relaxed) >= 0) return;
    char* dst = static_cast<char*>(rl->dsts[i]);
    int64_t got = 0;
    const int64_t want = rl->nbytes[i];
    while (got < want) {
      const int64_t piece = want - got < PIECE ? want - got : PIECE;
      const ssize_t k = WRITE ? pwrite(rl->fds[i], dst + got, (size_t)piece, (off_t)(rl->offsets[i] + got))
                              : pread(rl->fds[i], dst + got, (size_t)piece, (off_t)(rl->offsets[i] + got));
      if (k > 0) {
        got += k;
        continue;
      }
      if (k < 0 && errno == EINTR) continue;
      int expected = -1;
      if (rl->failed_run.compare_exchange_strong(expected, i)) rl->failed_errno.store(k < 0 ? errno : 0);
      return;
    }
  }
}

}  // namespace

namespace {
template <bool WRITE>
int run_all(const int32_t* fds, const int64_t* offsets, const int64_t* nbytes, void* const* bufs, int n, int n_threads,
            int64_t* failed) {
  if (n < 0 || n_threads < 1 || n_threads > 64 || (n > 0 && (!fds || !offsets || !nbytes || !bufs))) return MG_EINVAL;
  for (int i = 0; i < n; ++i)
    if (fds[i] < 0 || offsets[i] < 0 || nbytes[i] < 0 || (nbytes[i] > 0 && !bufs[i])) return MG_EINVAL;
  run_list rl;
  rl.fds = fds, rl.offsets = offsets, rl.nbytes = nbytes, rl.dsts = bufs, rl.n = n;
  const int helpers = (n_threads < n ? n_threads : n) - 1;
  std::vector<std::thread> pool;
  pool.reserve(helpers > 0 ? helpers : 0);
  for (int t = 0; t < helpers; ++t) pool.emplace_back(run_worker<WRITE>, &rl);
  run_worker<WRITE>(&rl);  // the calling thread takes runs too
  for (auto& th : pool) th.join();
  const int bad = rl.failed_run.load();
  if (bad >= 0) {
    if (failed) failed[0] = bad, failed[1] = rl.failed_errno.load();
    return MG_EIO;
  }
  return MG_OK;
}
}  // namespace

extern "C" int mg_host_read_runs(const int32_t* fds, const int64_t* offsets, const int64_t* nbytes, void* const* dsts, int n,
                                 int n_threads, int64_t* failed) {
  return run_all<false>(fds, offsets, nbytes, dsts, n, n_threads, failed);
}

extern "C" int mg_host_write_runs(const int32_t* fds, const int64_t* offsets, const int64_t* nbytes, void* const* srcs, int n,
                                  int n_threads, int64_t* failed) {
  return run_all<true>(fds, offsets, nbytes, srcs, n, n_threads, failed);
}
