// shine_torch_ext.cpp — Tier A's autograd nodes in C++ ("bound through a torch C++ extension").
//
// The unchanged drivers' loop body (shine_batch.py:115-210, shine_incre.py:118-181) at the reference's batch size is HOST
// bound: ~25 Python-level operations per iteration around five small launches, a third of it torch.autograd.Function's Python
// machinery (apply, ctx, the engine's call back into Python under the GIL) and the ctypes marshalling of the C ABI's pointer
// arrays (profiles/r04_tier_a_bench.log: 0.31-0.54 ms per iteration across boxes against 0.025 ms of kernels).  This module
// holds the nodes of that loop as torch::autograd::Node subclasses that call the same C ABI (include/shine_hip.h) directly:
//
//   query_feature    FeatureOctree.query_feature (model/feature_octree.py:237-244): shine_forward [+ the decoder rider]
//   fused_sdf        Decoder.sdf on query_feature's untouched output (model/decoder.py:49-63): ONE node for interpolation +
//                    decoder whose backward is shine_plan_batch + shine_interp_sdf_backward (the fused step's EXT build)
//   grad_coord       get_gradient(coord, pred) (utils/tools.py:175-185): shine_forward's closed-form d pred / d coord; its
//                    backward leaves d loss / d g with the fused node (shared Link)
//   bce_loss         sdf_bce_loss (utils/loss.py:17-24): shine_bce_loss, loss and d loss / d pred in one launch
//   cal_regularization  FeatureOctree.cal_regularization (model/feature_octree.py:246-255): a node only while a gradient is live
//   adam_step        FusedAdam.step (utils/tools.py:57-83's Adam): shine_adam_step without Python-side pointer arrays
//
// Everything the nodes do not cover — a differentiable backward through the fused node, a driver that touches the feature
// tensor, CPU tensors — goes back to the Python nodes of autograd_ops.py through two registered callbacks: same results, the
// round-4 cost.  No kernels here: host code only, linked against libshine_hip.so.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/csrc/autograd/functions/utils.h>
#include <torch/csrc/autograd/saved_variable.h>

#include <atomic>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/shine_hip.h"

namespace {

using torch::autograd::Node;
using torch::autograd::SavedVariable;
using torch::autograd::variable_list;
using at::Tensor;

void check(int rc, const char* what) {
  if (rc != SHINE_OK) throw std::runtime_error(std::string(what) + ": " + shine_error_string(rc));
}

void* cur_stream(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

Tensor f32c(const Tensor& t) {  // (the kernels read float32 contiguous memory in place)
  return (t.scalar_type() == at::kFloat && t.is_contiguous()) ? t : t.contiguous().to(at::kFloat);
}

// Strong references to Python objects held from C++ and released under the GIL (a node may die on an autograd thread, or after
// the interpreter: then the reference is leaked rather than touched).
struct PyKeep {
  py::object obj;
  explicit PyKeep(py::object o) : obj(std::move(o)) {}
  ~PyKeep() {
    if (!obj) return;
    if (Py_IsInitialized()) {
      py::gil_scoped_acquire gil;
      obj = py::object();
    } else {
      obj.release();
    }
  }
};

// cal_regularization's value riding on query_feature's launch (include/shine_hip.h shine_reg_rider): the tensors the rider reads,
// set once per frame by the Python side (FeatureOctree._reg_rider), and the launch counter that stamps the rows
struct RegRiderState {
  std::vector<Tensor> last, imp, stamp;  // per level: features_last_frame, importance_weight (float32 contiguous), uint32 stamps
  Tensor acc;                            // float32[8]
  uint32_t epoch = 0;
};

// What a FeatureOctree's launches need, refreshed by the Python side whenever the tables or the configuration change
// (FeatureOctree._ext_state): the table handle, the scalar configuration, the row counts.  Every autograd node takes a SNAPSHOT
// (snapshot()): the handle / rows / configuration of ITS forward, a strong reference to the octree and its table object (the
// Python nodes these replace kept ctx.octree: `del octree` before backward() must not free the handle under the node), and the
// tables epoch of its forward — an update() between a forward and its backward (hash slots move, rows are appended) is refused
// in apply() instead of planning the batch on other tables than the forward saw.
struct TierAState {
  uintptr_t tables = 0;
  shine_step_config cfg;
  std::vector<int64_t> rows;
  int64_t py_id = 0;  // the octree's key in the Python-side registry (fallback callbacks)
  bool async_growth = false;
  int64_t epoch = 0;                                 // FeatureOctree._tables_epoch when this state was set
  std::shared_ptr<std::atomic<int64_t>> epoch_now;   // ... and now (the Python side stores every change: set_epoch)
  std::shared_ptr<PyKeep> weak;                      // (weakref(octree), weakref(its _DeviceTables)): the live state is held by
                                                     // the octree itself and must not keep it alive
  std::shared_ptr<PyKeep> keep;                      // snapshots only: (octree, tables), strong
  std::shared_ptr<RegRiderState> reg;                // null: query_feature does not evaluate the regulariser
  int L() const { return cfg.n_levels; }
  void set(uintptr_t handle, const std::string& cfg_bytes, std::vector<int64_t> r, int64_t id, bool async_, int64_t epoch_,
           py::object owner) {
    if (cfg_bytes.size() != sizeof(shine_step_config)) throw std::runtime_error("TierAState.set: shine_step_config size mismatch");
    tables = handle;
    std::memcpy(&cfg, cfg_bytes.data(), sizeof(cfg));
    rows = std::move(r);
    py_id = id;
    async_growth = async_;
    epoch = epoch_;
    if (!epoch_now) epoch_now = std::make_shared<std::atomic<int64_t>>(epoch_);
    epoch_now->store(epoch_);
    weak = std::make_shared<PyKeep>(std::move(owner));
    if ((int)rows.size() != cfg.n_levels) throw std::runtime_error("TierAState.set: one row count per featured level");
  }
  // (called with the GIL held: from the Python-facing entry points)
  std::shared_ptr<TierAState> snapshot() const {
    auto s = std::make_shared<TierAState>(*this);
    if (weak && weak->obj) {
      py::tuple w = weak->obj.cast<py::tuple>();
      py::list strong;
      for (auto ref : w) strong.append(ref());  // (None for an object that is already gone: nothing to keep)
      s->keep = std::make_shared<PyKeep>(std::move(strong));
    }
    return s;
  }
  void check_epoch(const char* who) const {
    if (epoch_now && epoch_now->load() != epoch)
      throw std::runtime_error(std::string(who) + ": the octree's tables changed (update()) between this node's forward and its "
                               "backward; run backward() before growing the octree, or query again");
  }
};

// d loss / d (d pred / d coord), handed from grad_coord's backward to the fused node's (autograd runs the former first: pred is
// its input)
struct Link {
  Tensor q;
};

// (never destroyed: the interpreter may be gone when static destructors run)
py::object* g_interp_backward = nullptr;  // (octree id, g, coord, need_coord, feats) -> (grad_coord, *grad_feats): OctreeInterpBackward
py::object* g_fused_split = nullptr;      // (octree id, g, coord, feats, mlp, needs) -> grads of (coord, *feats, *mlp): the split nodes
py::object* g_read_done = nullptr;        // (octree id): FeatureOctree._tables_read_done (asynchronous growth only)

std::vector<const float*> ptrs(const std::vector<Tensor>& ts) {
  std::vector<const float*> p;
  p.reserve(ts.size());
  for (const auto& t : ts) p.push_back(t.defined() ? t.data_ptr<float>() : nullptr);
  return p;
}

// persistent scratch, as ops._workspace / dp.plan_batch keep it: the fused step's partial sums per (device, stream), the plan's
// scratch per device (grow-only; neither is ever used inside a captured graph from here)
std::mutex g_mu;
std::map<std::pair<int, uintptr_t>, Tensor> g_step_ws;
std::map<int, Tensor> g_plan_ws;

Tensor step_workspace(const Tensor& like, const shine_step_config& cfg, void* stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto key = std::make_pair((int)like.get_device(), (uintptr_t)stream);
  auto it = g_step_ws.find(key);
  if (it != g_step_ws.end()) return it->second;
  const size_t bytes = shine_train_step_workspace_bytes(&cfg, -1);
  Tensor ws = at::empty({(int64_t)(bytes ? bytes : 1)}, like.options().dtype(at::kByte));
  g_step_ws[key] = ws;
  return ws;
}

// ---------------------------------------------------------------------------------------------------------------- nodes

// feat = query_feature(coord): the backward (a driver that consumed the features itself instead of handing them to Decoder.sdf)
// is the Python node's — differentiable twice, as the eikonal configurations need it
struct InterpNode : public Node {
  std::string name() const override { return "OctreeInterp[ext]"; }
  SavedVariable coord;
  std::vector<SavedVariable> feats;
  int64_t py_id = 0;
  bool need_coord = false;
  std::shared_ptr<TierAState> st;  // snapshot: keeps the octree (the registry behind py_id holds weak references) and the epoch
  variable_list apply(variable_list&& grads) override {
    st->check_epoch("query_feature's backward");
    py::gil_scoped_acquire gil;
    if (!grads[0].defined()) return variable_list(1 + feats.size());
    py::list fl;
    for (auto& f : feats) fl.append(f.unpack());
    py::tuple out = (*g_interp_backward)(py_id, grads[0], coord.unpack(), need_coord, fl);
    variable_list res;
    for (size_t i = 0; i < out.size(); ++i) res.push_back(out[i].is_none() ? Tensor() : out[i].cast<Tensor>());
    return res;
  }
  void release_variables() override {
    coord.reset_data();
    for (auto& f : feats) f.reset_data();
  }
};

struct FusedSdfNode : public Node {
  std::string name() const override { return "FusedInterpSdf[ext]"; }
  SavedVariable coord;
  std::vector<SavedVariable> feats, mlp;
  std::shared_ptr<TierAState> st;
  std::shared_ptr<Link> link;
  std::vector<bool> need;  // coord, feats..., mlp...
  bool deterministic = false;

  variable_list apply(variable_list&& grads) override {
    const int L = st->L();
    Tensor g = grads[0];
    Tensor q = link->q;
    link->q = Tensor();
    variable_list out(1 + L + 6);
    if (!g.defined() && !q.defined()) return out;
    st->check_epoch("the fused query_feature -> sdf node's backward");
    Tensor c = coord.unpack();
    std::vector<Tensor> F, M;
    for (auto& f : feats) F.push_back(f.unpack());
    for (auto& m : mlp) M.push_back(m.unpack());
    if (at::GradMode::is_enabled()) {  // a differentiable backward: the split, twice-differentiable Python nodes
      py::gil_scoped_acquire gil;
      py::list fl, ml, nl;
      for (auto& f : F) fl.append(f);
      for (auto& m : M) ml.append(m);
      for (bool b : need) nl.append(b);
      py::tuple res = (*g_fused_split)(st->py_id, g, c, fl, ml, nl);
      for (size_t i = 0; i < res.size() && i < out.size(); ++i) out[i] = res[i].is_none() ? Tensor() : res[i].cast<Tensor>();
      return out;
    }
    at::NoGradGuard ng;
    Tensor cc = f32c(c.detach());
    const int64_t n = cc.size(0);
    void* stream = cur_stream(cc);
    if (!g.defined()) g = at::zeros({n}, cc.options());
    g = f32c(g);
    bool need_m = false;
    for (int k = 0; k < 6; ++k) need_m = need_m || need[1 + L + k];
    // every gradient of this node as a view of ONE flat buffer, which the plan's first pass clears
    std::vector<int64_t> sizes;
    int64_t total = 0;
    for (int s = 0; s < L; ++s) sizes.push_back(need[1 + s] ? F[s].numel() : 0);
    for (int k = 0; k < 6; ++k) sizes.push_back(need_m ? M[k].numel() : 0);
    for (auto v : sizes) total += v;
    Tensor flat = at::empty({(total + 3) / 4 * 4}, cc.options());
    // --- shine_plan_batch (dp.plan_batch)
    shine_step_config cfg = st->cfg;
    const shine_tables* t = reinterpret_cast<const shine_tables*>(st->tables);
    Tensor perm = at::empty({n}, cc.options().dtype(at::kInt)), slots = at::empty({n, (int64_t)L}, cc.options().dtype(at::kInt));
    {
      size_t need_bytes = 0;
      check(shine_plan_batch(t, &cfg, nullptr, n, nullptr, nullptr, nullptr, 0, nullptr, &need_bytes, stream), "shine_plan_batch");
      Tensor ws;
      {
        std::lock_guard<std::mutex> lk(g_mu);
        Tensor& ent = g_plan_ws[(int)cc.get_device()];
        if (!ent.defined() || (size_t)ent.numel() < need_bytes)
          ent = at::empty({(int64_t)(need_bytes + need_bytes / 4 + 1)}, cc.options().dtype(at::kByte));
        ws = ent;
      }
      size_t have = (size_t)ws.numel();
      check(shine_plan_batch(t, &cfg, cc.data_ptr<float>(), n, perm.data_ptr<int>(), slots.data_ptr<int>(), flat.data_ptr<float>(),
                             (size_t)flat.numel() * sizeof(float), ws.data_ptr(), &have, stream),
            "shine_plan_batch");
      if (st->async_growth) {
        py::gil_scoped_acquire gil;
        (*g_read_done)(st->py_id);
      }
    }
    // (each view goes straight into `out` and is referenced from nowhere else: AccumulateGrad then adopts it as .grad instead
    // of cloning it — one small copy launch per parameter otherwise)
    int64_t off = 0;
    for (size_t i = 0; i < sizes.size(); ++i) {
      const Tensor& p = i < (size_t)L ? F[i] : M[i - L];
      if (sizes[i]) out[1 + i] = flat.narrow(0, off, sizes[i]).view(p.sizes());
      off += sizes[i];
    }
    std::vector<Tensor> Fc, Mc;
    for (auto& f : F) Fc.push_back(f32c(f));
    for (auto& m : M) Mc.push_back(f32c(m));
    cfg.sorted_input = 1;
    cfg.decoder_grad_on = need_m ? 1 : 0;
    cfg.kernel_variant |= deterministic ? 0x4000 : 0;  // (OR: FeatureOctree.DEBUG_VARIANT_BITS travel in st->cfg)
    Tensor ws = step_workspace(cc, cfg, stream);
    auto fp = ptrs(Fc), mp = ptrs(Mc);
    std::vector<float*> gf, gm;
    for (int s = 0; s < L; ++s) gf.push_back(out[1 + s].defined() ? out[1 + s].data_ptr<float>() : nullptr);
    for (int k = 0; k < 6; ++k) gm.push_back(out[1 + L + k].defined() ? out[1 + L + k].data_ptr<float>() : nullptr);
    Tensor qc = q.defined() ? f32c(q) : Tensor();
    check(shine_interp_sdf_backward(t, &cfg, cc.data_ptr<float>(), perm.data_ptr<int>(), slots.data_ptr<int>(), g.data_ptr<float>(),
                                    qc.defined() ? qc.data_ptr<float>() : nullptr, n, fp.data(), st->rows.data(), mp.data(),
                                    gf.data(), need_m ? gm.data() : nullptr, ws.data_ptr(), (size_t)ws.numel(), stream),
          "shine_interp_sdf_backward");
    flat.reset();
    return out;
  }
  void release_variables() override {
    coord.reset_data();
    for (auto& f : feats) f.reset_data();
    for (auto& m : mlp) m.reset_data();
  }
};

struct GradCoordNode : public Node {
  std::string name() const override { return "InterpSdfGradCoord[ext]"; }
  std::shared_ptr<Link> link;
  size_t n_in = 0;
  variable_list apply(variable_list&& grads) override {
    if (at::GradMode::is_enabled())
      throw std::runtime_error("get_gradient's node (d pred / d coord in closed form) is differentiable once");
    if (grads[0].defined()) {
      Tensor q = f32c(grads[0]);
      link->q = link->q.defined() ? link->q + q : q;
    }
    return variable_list(n_in);
  }
};

struct BceNode : public Node {
  std::string name() const override { return "SdfBce[ext]"; }
  Tensor dpred;
  variable_list apply(variable_list&& grads) override {
    variable_list out(1);
    if (grads[0].defined()) out[0] = dpred * grads[0];
    return out;
  }
  void release_variables() override { dpred.reset(); }
};

// reg = cal_regularization() (model/feature_octree.py:246-255; autograd_ops.OctreeRegularizer): the rows the octree's last
// query addressed are flagged (shine_mark_touched) and summed by one row-parallel launch that clears the flags again.  A node
// exists only while a level's features_last_frame is still a detached copy (`live`): from the second frame on the reference holds
// an attached clone and d reg / d F cancels (:160) — then the value is a constant of the graph and nothing runs in backward.
void mark_query_rows(const TierAState& st, const Tensor& c, const std::vector<Tensor>& flags, void* stream) {
  std::vector<unsigned char*> fl;
  for (auto& f : flags) fl.push_back(f.data_ptr<unsigned char>());
  check(shine_mark_touched(reinterpret_cast<const shine_tables*>(st.tables), &st.cfg, c.data_ptr<float>(), nullptr, nullptr,
                           c.size(0), st.rows.data(), fl.data(), stream),
        "shine_mark_touched");
}

struct RegNode : public Node {
  std::string name() const override { return "OctreeRegularizer[ext]"; }
  std::shared_ptr<TierAState> st;
  Tensor coord;                        // detached, float32 contiguous
  std::vector<SavedVariable> feats;
  std::vector<Tensor> last, imp, flags;  // detached
  std::vector<bool> on;                // per level: the gradient is live and wanted
  variable_list apply(variable_list&& grads) override {
    const int L = st->L();
    variable_list out(L);
    if (!grads[0].defined()) return out;
    if (at::GradMode::is_enabled()) throw std::runtime_error("cal_regularization's node is differentiable once");
    st->check_epoch("cal_regularization's backward");
    at::NoGradGuard ng;
    std::vector<Tensor> F;
    for (auto& f : feats) F.push_back(f32c(f.unpack()));
    void* stream = cur_stream(coord);
    mark_query_rows(*st, coord, flags, stream);  // (forward's launch cleared the flags)
    std::vector<float*> gp;
    std::vector<int32_t> gon;
    std::vector<unsigned char*> fl;
    for (int s = 0; s < L; ++s) {
      if (on[s]) out[s] = at::zeros_like(F[s]);
      gp.push_back(on[s] ? out[s].data_ptr<float>() : nullptr);
      gon.push_back(on[s] ? 1 : 0);
      fl.push_back(flags[s].data_ptr<unsigned char>());
    }
    Tensor scratch = at::empty({1}, coord.options().dtype(at::kDouble));
    auto fp = ptrs(F), lp = ptrs(last), ip = ptrs(imp);
    check(shine_regularize(L, fp.data(), lp.data(), ip.data(), gp.data(), fl.data(), st->rows.data(), gon.data(), 1.0f,
                           scratch.data_ptr<double>(), 0, 0, stream),
          "shine_regularize");
    Tensor g = grads[0].to(at::kFloat);
    for (int s = 0; s < L; ++s)
      if (on[s]) out[s].mul_(g);
    return out;
  }
  void release_variables() override {
    for (auto& f : feats) f.reset_data();
  }
};

bool requires_any(const Tensor& c, const std::vector<Tensor>& a, const std::vector<Tensor>& b = {}) {
  if (c.defined() && c.requires_grad()) return true;
  for (auto& t : a)
    if (t.requires_grad()) return true;
  for (auto& t : b)
    if (t.requires_grad()) return true;
  return false;
}

template <class N>
std::shared_ptr<N> make_node(const variable_list& inputs) {
  auto node = std::shared_ptr<N>(new N(), torch::autograd::deleteNode);
  node->set_next_edges(torch::autograd::collect_next_edges(inputs));
  return node;
}

// ---------------------------------------------------------------------------------------------------------------- calls

// -> (feat [n, 8], pred [n] or None).  mlp: empty, or the six tensors of the decoder whose output rides on the launch.
// reg (third result): with st->reg set, the regulariser of THIS query as a 0-dim view of the rider's accumulator ring — valid until
// seven more queries of the octree have run (FeatureOctree.cal_regularization clones it)
std::tuple<Tensor, c10::optional<Tensor>, c10::optional<Tensor>> query_feature(const std::shared_ptr<TierAState>& st,
                                                                              const Tensor& coord, const std::vector<Tensor>& feats,
                                                                              const std::vector<Tensor>& mlp) {
  const int L = st->L();
  TORCH_CHECK((int)feats.size() == L, "query_feature: one table per featured level");
  TORCH_CHECK(coord.is_cuda() && coord.scalar_type() == at::kFloat && coord.dim() == 2 && coord.size(1) == 3,
              "coord must be a CUDA float32 tensor of shape [N,3]");
  Tensor feat, pred, reg;
  {
    at::AutoDispatchBelowADInplaceOrView guard;
    Tensor c = coord.detach().contiguous();
    const int64_t n = c.size(0);
    void* stream = cur_stream(c);
    feat = at::empty({n, 8}, c.options());
    std::vector<Tensor> Fc, Mc;
    for (auto& f : feats) Fc.push_back(f32c(f.detach()));
    for (auto& m : mlp) Mc.push_back(f32c(m.detach()));
    if (!Mc.empty()) pred = at::empty({n}, c.options());
    auto fp = ptrs(Fc), mp = ptrs(Mc);
    shine_step_config cfg = st->cfg;
    shine_reg_rider rider;
    if (st->reg && n > 0) {
      RegRiderState& rs = *st->reg;
      std::memset(&rider, 0, sizeof(rider));
      for (int s = 0; s < L; ++s) {
        rider.last[s] = rs.last[s].data_ptr<float>();
        rider.imp[s] = rs.imp[s].data_ptr<float>();
        rider.stamp[s] = reinterpret_cast<uint32_t*>(rs.stamp[s].data_ptr<int32_t>());
      }
      rider.acc = rs.acc.data_ptr<float>();
      rider.epoch = ++rs.epoch;
      cfg.reg_rider = &rider;
      reg = rs.acc.select(0, (int64_t)(rider.epoch & 7u));
    }
    if (n > 0)
      check(shine_forward(reinterpret_cast<const shine_tables*>(st->tables), &cfg, c.data_ptr<float>(), n, fp.data(),
                          st->rows.data(), Mc.empty() ? nullptr : mp.data(), feat.data_ptr<float>(),
                          pred.defined() ? pred.data_ptr<float>() : nullptr, nullptr, nullptr, stream),
            "shine_forward");
  }
  if (at::GradMode::is_enabled() && requires_any(coord, feats)) {
    variable_list inputs{coord};
    inputs.insert(inputs.end(), feats.begin(), feats.end());
    auto node = make_node<InterpNode>(inputs);
    node->coord = SavedVariable(coord, false);
    for (auto& f : feats) node->feats.emplace_back(f, false);
    node->py_id = st->py_id;
    node->st = st->snapshot();
    node->need_coord = coord.requires_grad();
    torch::autograd::create_gradient_edge(feat, node);
  }
  return {feat, pred.defined() ? c10::optional<Tensor>(pred) : c10::nullopt, reg.defined() ? c10::optional<Tensor>(reg) : c10::nullopt};
}

// pred = Decoder.sdf(feature) for the untouched feature of query_feature(coord) -> (pred, link)
std::pair<Tensor, std::shared_ptr<Link>> fused_sdf(const std::shared_ptr<TierAState>& st, const Tensor& feat, const Tensor& coord,
                                                   const c10::optional<Tensor>& spec_pred, const std::vector<Tensor>& feats,
                                                   const std::vector<Tensor>& mlp, bool deterministic) {
  const int L = st->L();
  TORCH_CHECK((int)feats.size() == L && mlp.size() == 6, "fused_sdf: L feature tables and six decoder tensors");
  Tensor pred;
  if (spec_pred.has_value()) {
    pred = spec_pred->detach();
  } else {
    at::AutoDispatchBelowADInplaceOrView guard;
    Tensor f = f32c(feat.detach());
    std::vector<Tensor> Mc;
    for (auto& m : mlp) Mc.push_back(f32c(m.detach()));
    auto mp = ptrs(Mc);
    pred = at::empty({f.size(0)}, f.options());
    check(shine_mlp_forward(f.data_ptr<float>(), f.size(0), mp.data(), pred.data_ptr<float>(), cur_stream(f)), "shine_mlp_forward");
  }
  auto link = std::make_shared<Link>();
  variable_list inputs{coord};
  inputs.insert(inputs.end(), feats.begin(), feats.end());
  inputs.insert(inputs.end(), mlp.begin(), mlp.end());
  auto node = make_node<FusedSdfNode>(inputs);
  node->coord = SavedVariable(coord, false);
  for (auto& f : feats) node->feats.emplace_back(f, false);
  for (auto& m : mlp) node->mlp.emplace_back(m, false);
  node->st = st->snapshot();
  node->link = link;
  node->deterministic = deterministic;
  for (auto& t : inputs) node->need.push_back(t.requires_grad());
  torch::autograd::create_gradient_edge(pred, node);
  return {pred, link};
}

// raw = d pred / d coord [n, 3] (get_gradient without the sigma factor the drivers multiply in afterwards)
Tensor grad_coord(const std::shared_ptr<TierAState>& st, const Tensor& pred, const Tensor& coord, const std::shared_ptr<Link>& link,
                  const std::vector<Tensor>& feats, const std::vector<Tensor>& mlp) {
  Tensor raw;
  {
    at::AutoDispatchBelowADInplaceOrView guard;
    Tensor c = coord.detach().contiguous();
    const int64_t n = c.size(0);
    std::vector<Tensor> Fc, Mc;
    for (auto& f : feats) Fc.push_back(f32c(f.detach()));
    for (auto& m : mlp) Mc.push_back(f32c(m.detach()));
    auto fp = ptrs(Fc), mp = ptrs(Mc);
    raw = at::empty({n, 3}, c.options());
    shine_step_config cfg = st->cfg;
    cfg.sigma = 1.0f;
    check(shine_forward(reinterpret_cast<const shine_tables*>(st->tables), &cfg, c.data_ptr<float>(), n, fp.data(), st->rows.data(),
                        mp.data(), nullptr, nullptr, nullptr, raw.data_ptr<float>(), cur_stream(c)),
          "shine_forward");
  }
  variable_list inputs{pred, coord};
  inputs.insert(inputs.end(), feats.begin(), feats.end());
  inputs.insert(inputs.end(), mlp.begin(), mlp.end());
  auto node = make_node<GradCoordNode>(inputs);
  node->link = link;
  node->n_in = inputs.size();
  torch::autograd::create_gradient_edge(raw, node);
  return raw;
}

Tensor bce_loss(const Tensor& pred, const Tensor& label, const c10::optional<Tensor>& weight, double sigma, bool reduction_sum) {
  Tensor loss, dpred;
  {
    at::AutoDispatchBelowADInplaceOrView guard;
    Tensor p = f32c(pred.detach()), l = f32c(label.detach());
    Tensor w = weight.has_value() ? f32c(weight->detach()) : Tensor();
    const int64_t n = p.size(0);
    Tensor out = at::empty({n + 1}, p.options());  // [d loss / d pred (n) | loss]
    check(shine_bce_loss(p.data_ptr<float>(), l.data_ptr<float>(), w.defined() ? w.data_ptr<float>() : nullptr, n, (float)sigma,
                         reduction_sum ? 1 : 0, out.data_ptr<float>() + n, out.data_ptr<float>(), cur_stream(p)),
          "shine_bce_loss");
    loss = out.select(0, n);
    dpred = out.narrow(0, 0, n);
  }
  if (at::GradMode::is_enabled() && pred.requires_grad()) {
    auto node = make_node<BceNode>({pred});
    node->dpred = dpred;
    torch::autograd::create_gradient_edge(loss, node);
  }
  return loss;
}

// FeatureOctree.cal_regularization for the coordinates of the octree's last query.  flags: one uint8 per row and level, all zero
// (left zero again).  live[s]: level s's features_last_frame is a detached copy (its gradient does not cancel).
Tensor cal_regularization(const std::shared_ptr<TierAState>& st, const Tensor& coord, const std::vector<Tensor>& feats,
                          const std::vector<Tensor>& last, const std::vector<Tensor>& imp, const std::vector<Tensor>& flags,
                          const std::vector<bool>& live) {
  const int L = st->L();
  TORCH_CHECK((int)feats.size() == L && (int)last.size() == L && (int)imp.size() == L && (int)flags.size() == L && (int)live.size() == L,
              "cal_regularization: one tensor per featured level");
  TORCH_CHECK(coord.is_cuda() && coord.dim() == 2 && coord.size(1) == 3, "coord must be a CUDA tensor of shape [N,3]");
  Tensor reg;
  std::vector<bool> on(L, false);
  bool need_grad = false;
  for (int s = 0; s < L; ++s) {
    on[s] = live[s] && at::GradMode::is_enabled() && feats[s].requires_grad();
    need_grad = need_grad || on[s];
  }
  Tensor c;
  std::vector<Tensor> lastc, impc;
  {
    at::AutoDispatchBelowADInplaceOrView guard;
    c = f32c(coord.detach());
    void* stream = cur_stream(c);
    std::vector<Tensor> F;
    for (int s = 0; s < L; ++s) {
      TORCH_CHECK(flags[s].scalar_type() == at::kByte && flags[s].is_contiguous() && flags[s].numel() >= st->rows[s] + 1,
                  "cal_regularization: one uint8 flag per row");
      F.push_back(f32c(feats[s].detach()));
      lastc.push_back(f32c(last[s].detach()));
      impc.push_back(f32c(imp[s].detach()));
    }
    mark_query_rows(*st, c, flags, stream);
    Tensor out = at::zeros({1}, c.options().dtype(at::kDouble));
    std::vector<float*> gp(L, nullptr);
    std::vector<int32_t> gon(L, 0);
    std::vector<unsigned char*> fl;
    for (auto& f : flags) fl.push_back(f.data_ptr<unsigned char>());
    auto fp = ptrs(F), lp = ptrs(lastc), ip = ptrs(impc);
    check(shine_regularize(L, fp.data(), lp.data(), ip.data(), gp.data(), fl.data(), st->rows.data(), gon.data(), 0.0f,
                           out.data_ptr<double>(), 1, 0, stream),
          "shine_regularize");
    reg = out.select(0, 0).to(at::kFloat);
  }
  if (need_grad) {
    auto node = make_node<RegNode>(feats);
    node->st = st->snapshot();
    node->coord = c;
    for (auto& f : feats) node->feats.emplace_back(f, false);
    node->last = lastc;
    node->imp = impc;
    node->flags = flags;
    node->on = on;
    torch::autograd::create_gradient_edge(reg, node);
  }
  return reg;
}

void adam_step(const std::vector<Tensor>& params, const std::vector<Tensor>& grads, const std::vector<Tensor>& m,
               const std::vector<Tensor>& v, const std::vector<double>& lr, const std::vector<double>& wd, double b1, double b2,
               double eps, int64_t step, bool zero_grad, const std::vector<c10::optional<Tensor>>& flags) {
  const size_t n = params.size();
  TORCH_CHECK(n > 0 && n <= 16 && grads.size() == n && m.size() == n && v.size() == n && lr.size() == n && wd.size() == n,
              "adam_step: up to 16 tensors, one entry each");
  std::vector<float*> pp, gp, mp, vp;
  std::vector<int64_t> numel;
  std::vector<float> lrf, wdf;
  std::vector<unsigned char*> fl;
  bool any_flag = false;
  for (size_t i = 0; i < n; ++i) {
    TORCH_CHECK(params[i].is_cuda() && params[i].scalar_type() == at::kFloat && params[i].is_contiguous() && grads[i].is_contiguous(),
                "FusedAdam needs contiguous CUDA float32 parameters and grads");
    pp.push_back(params[i].data_ptr<float>());
    gp.push_back(grads[i].data_ptr<float>());
    mp.push_back(m[i].data_ptr<float>());
    vp.push_back(v[i].data_ptr<float>());
    numel.push_back(params[i].numel());
    lrf.push_back((float)lr[i]);
    wdf.push_back((float)wd[i]);
    unsigned char* f = nullptr;
    if (i < flags.size() && flags[i].has_value()) {
      f = flags[i]->data_ptr<unsigned char>();
      any_flag = true;
    }
    fl.push_back(f);
  }
  check(shine_adam_step((int32_t)n, pp.data(), gp.data(), mp.data(), vp.data(), numel.data(), lrf.data(), wdf.data(), (float)b1,
                        (float)b2, (float)eps, step, zero_grad ? 1 : 0, any_flag ? fl.data() : nullptr, cur_stream(params[0])),
        "shine_adam_step");
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "Tier A autograd nodes of shine_mapping_amd in C++ (shine_torch_ext.cpp)";
  py::class_<TierAState, std::shared_ptr<TierAState>>(m, "TierAState")
      .def(py::init<>())
      .def("set", [](TierAState& s, uintptr_t handle, py::bytes cfg, std::vector<int64_t> rows, int64_t id, bool async_,
                     int64_t epoch, py::object owner) {
        s.set(handle, std::string(cfg), std::move(rows), id, async_, epoch, std::move(owner));
      })
      .def("set_epoch", [](TierAState& s, int64_t epoch) {  // FeatureOctree._tables_epoch's setter
        if (s.epoch_now) s.epoch_now->store(epoch);
      })
      // cal_regularization's value rides on query_feature (RegRiderState); empty lists switch it off
      .def("set_reg", [](TierAState& s, std::vector<Tensor> last, std::vector<Tensor> imp, std::vector<Tensor> stamp, Tensor acc) {
        if (last.empty()) {
          s.reg.reset();
          return;
        }
        const size_t L = (size_t)s.L();
        TORCH_CHECK(last.size() == L && imp.size() == L && stamp.size() == L, "set_reg: one tensor per featured level");
        TORCH_CHECK(acc.is_cuda() && acc.scalar_type() == at::kFloat && acc.numel() == 8, "set_reg: acc is a CUDA float32[8]");
        for (size_t k = 0; k < L; ++k) {
          TORCH_CHECK(last[k].is_cuda() && last[k].scalar_type() == at::kFloat && last[k].is_contiguous() &&
                      imp[k].is_cuda() && imp[k].scalar_type() == at::kFloat && imp[k].is_contiguous() &&
                      last[k].size(0) == s.rows[k] + 1 && imp[k].size(0) == s.rows[k] + 1,
                      "set_reg: features_last_frame / importance_weight as contiguous CUDA float32 [rows + 1, 8]");
          TORCH_CHECK(stamp[k].is_cuda() && stamp[k].scalar_type() == at::kInt && stamp[k].numel() >= s.rows[k] + 1,
                      "set_reg: one int32 stamp per row");
        }
        auto keep_epoch = s.reg ? s.reg->epoch : 0u;  // (stamps may be re-used: the epoch never goes back)
        s.reg = std::make_shared<RegRiderState>();
        s.reg->last = std::move(last), s.reg->imp = std::move(imp), s.reg->stamp = std::move(stamp), s.reg->acc = std::move(acc);
        s.reg->epoch = keep_epoch;
      });
  py::class_<Link, std::shared_ptr<Link>>(m, "Link").def("pending", [](Link& l) { return l.q.defined(); });
  m.def("set_callbacks", [](py::object interp_backward, py::object fused_split, py::object read_done) {
    g_interp_backward = new py::object(std::move(interp_backward));
    g_fused_split = new py::object(std::move(fused_split));
    g_read_done = new py::object(std::move(read_done));
  });
  m.def("query_feature", &query_feature);
  m.def("fused_sdf", &fused_sdf);
  m.def("grad_coord", &grad_coord);
  m.def("bce_loss", &bce_loss);
  m.def("cal_regularization", &cal_regularization);
  m.def("adam_step", &adam_step);
  m.def("config_bytes", []() { return (int64_t)sizeof(shine_step_config); });
}
