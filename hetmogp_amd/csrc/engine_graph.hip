// engine_graph.hip -- hipGraph capture / replay of one small-model evaluation (hmogp_engine::graph_step and its bookkeeping).
// Split out of engine.hip in round 6 (no behaviour change); declarations: engine_impl.h.
#include "engine_impl.h"

void hmogp_engine::drop_graphs(bool keep_warm) {
  for (auto& g : graphs) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
  }
  graphs.clear();
  if (!keep_warm) warm_keys.clear();
}

std::vector<long long> hmogp_engine::graph_key(const hmogp_params* p) const {
  std::vector<long long> k{(long long)p->group_mask, (!p->m_u && !p->L_flat) ? 1 : 0};
  for (int t = 0; t < T; ++t) k.push_back(p->row_begin ? p->row_begin[t] : 0), k.push_back(p->row_end ? p->row_end[t] : tasks[t].N);
  for (int q = 0; q < Q; ++q) k.push_back(p->forced_rung ? p->forced_rung[q] : -2);
  return k;
}

bool hmogp_engine::graph_step(const hmogp_params* p, hmogp_outputs* out) {
  static const bool enabled = [] {   // HMOGP_SMALL_GRAPH=0: no graphs (A/B runs)
    const char* e = getenv("HMOGP_SMALL_GRAPH");
    return !(e && e[0] == '0');
  }();
  if (!enabled || !p || !out || out->dL_dS || small_veto || comm) return false;
  HIP_TRY(hipSetDevice(device));
  sharded_call = false;   // (only plain hmogp_elbo_grad comes here; a stale `true` from an earlier hmogp_step_begin / sharded step made
                          //  THIS decision drop the small path and re-wire the streams for one call: ADVICE r5)
  decide_mode(p);
  if (!small_path) return false;
  const std::vector<long long> key = graph_key(p);
  SmallGraph* hit = nullptr;
  for (auto& g : graphs)
    if (g.key == key) hit = &g;
  if (!hit) {
    if (std::find(warm_keys.begin(), warm_keys.end(), key) == warm_keys.end()) {
      pending_warm = key;          // first sight: a normal evaluation sizes every workspace; warm once it has SUCCEEDED
      return false;
    }
    if (graphs.size() >= 32) drop_graphs();
    // ---- capture: the normal code path, recorded instead of executed ------------------------------------------------------
    SmallGraph g;
    g.key = key;
    HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    bool ok = true;
    std::string why;
    try {
      begin(p, false);
      if (!out) throw EngineError{HMOGP_E_INVALID, "null outputs"};
      finish_enqueue(out);
    } catch (const EngineError& e) {
      ok = false, why = e.msg;
    } catch (const HipError& e) {
      ok = false, why = hipGetErrorString(e.code);
    }
    const hipError_t ec = hipStreamEndCapture(st, &g.graph);
    if (!ok || ec != hipSuccess || !g.graph) {
      if (g.graph) (void)hipGraphDestroy(g.graph);
      (void)hipGetLastError();
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)
        throw EngineError{HMOGP_E_NO_DEVICE, "a failed hipGraph capture left the engine's stream in capture mode: " + why};
      (void)hipGetLastError();
      began = false;
      warm_keys.clear();            // (do not try again for this engine's current keys; the normal path reports real errors)
      graphs_broken = true;
      return false;
    }
    if (hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) != hipSuccess) {
      (void)hipGraphDestroy(g.graph);
      (void)hipGetLastError();
      began = false;
      graphs_broken = true;
      return false;
    }
    graphs.push_back(g);
    hit = &graphs.back();
    ++graph_captures;
  } else {
    // ---- replay: only the HOST side of begin() (validation, parameter image, pool plan, output layout) ----------------------
    began = false, exchanged = false;
    spans.clear(), pool_used = 0;
    for (int c = 0; c < NCAT; ++c) ms[c] = 0.0, launches[c] = 0;
    upload_params(p, false);
    plan_pools();
    kuu_key_valid = false;
    for (int q = 0; q < Q; ++q)
      if (rung[q] == -2) rung[q] = -1;
    small_info_pending = true;
    info_early = false;            // (the captured evaluation delivers its info words with the results)
    began = true;
    fin_layout(out);
    ++graph_replays;
  }
  HIP_TRY(hipGraphLaunch(hit->exec, st));
  via_graph = true;
  try {
    finish_tail(out);
  } catch (...) {
    via_graph = false;
    throw;
  }
  via_graph = false;
  return true;
}

void hmogp_engine::mark_warm() {
  if (!pending_warm.empty() && small_path && warm_keys.size() < 64) warm_keys.push_back(pending_warm);
  pending_warm.clear();
}
