// Host-only planning of the cascade launches of one predictor pass: which side stream every launch goes to, and in which
// order the launches are issued.  Pure functions (no HIP): run_predict (host.hip) calls them, tests call them through
// sacamd_plan_cascade_streams without a GPU.
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>

namespace sacamd {

// One cascade launch as the planner sees it: the group it belongs to (a group waits for its own OLS classes) and its work
// (sum over its items of taps x samples; 1e300 marks the whole-CU layout, which is issued first).
struct CascadePlanIn { int group; double work; };

// Launches on one stream run one after the other, and a launch lasts as long as its slowest item whatever its size.
//  * order: by group, then by descending work (the whole-CU layout first: its workgroups need drained CUs);
//  * a throughput-bound launch (work >= small_w) goes to the least loaded stream of its group's pool -- when a group has more
//    launches than streams the later ones queue behind the SMALLEST of the big ones (round 4: round-robin put a 0.4-s launch
//    behind the 8-s one and the generation ended 0.3-0.8 s late);
//  * a small launch (work < small_w: about one item's latency at the chip's rate) prefers a stream of the pool that holds no
//    throughput-bound launch -- all of those end together, when the chip drains -- and small launches count as small_w each,
//    i.e. they are balanced by their number, not by their taps (round 6: four launches of 1-5 items, 6-14 s each, had queued
//    on one stream and ended 1.6 s after everything else: profiles/r06/launch_trace_1536_before_streams.txt).
// pool[g] = streams group g may use (non-empty).  Returns false when a launch names a group without streams.
inline bool plan_cascade_streams(const std::vector<CascadePlanIn> &L, const std::vector<std::vector<int>> &pool, double small_w,
                                 std::vector<size_t> &order, std::vector<int> &stream) {
  const size_t n = L.size();
  order.resize(n); stream.assign(n, -1);
  std::iota(order.begin(), order.end(), (size_t)0);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
    return L[a].group != L[b].group ? L[a].group < L[b].group : L[a].work > L[b].work; });
  int nstreams = 0;
  for (const auto &p : pool) for (int si : p) nstreams = std::max(nstreams, si + 1);
  std::vector<double> load(nstreams, 0.0);
  std::vector<int> nbig(nstreams, 0);
  for (size_t q : order) {
    const int g = L[q].group;
    if (g < 0 || g >= (int)pool.size() || pool[g].empty()) return false;
    const bool big = L[q].work >= small_w;
    auto better = [&](int a, int b) {
      if (!big && (nbig[a] > 0) != (nbig[b] > 0)) return nbig[a] == 0;
      return load[a] < load[b]; };
    int best = pool[g][0];
    for (int si : pool[g]) if (better(si, best)) best = si;
    stream[q] = best;
    load[best] += big ? std::min(L[q].work, 1e290) + 1.0 : small_w;
    nbig[best] += big;
  }
  return true;
}

}  // namespace sacamd
