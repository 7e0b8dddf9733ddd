// graph_fix.hip — rewrite the memset nodes of a captured hipGraph into kernel nodes.
//
// Why: on ROCm 7.2 / MI355X a replayed graph that contains memset nodes (hipMemsetAsync captured
// from this library's callers — in practice MIOpen's conv-backward solvers, which zero their
// split-K accumulators that way) was observed to corrupt results: the memset is not reliably
// ordered against its neighbouring kernel nodes.  Kernel nodes are.  This entry
// point takes the raw hipGraph_t BEFORE instantiation (PyTorch: CUDAGraph(keep_graph=True) ->
// raw_cuda_graph()), and for every memset node adds a fill-kernel node with the same
// predecessors and successors, then destroys the memset node.  The caller instantiates afterwards.
#include <algorithm>
#include <vector>

#include "hd_common.h"

namespace {

// generic byte-granular 2-D fill (covers every hipMemsetParams shape); value holds the element pattern
__global__ __launch_bounds__(256) void graph_fill_bytes_kernel(unsigned char* __restrict__ dst, size_t pitch,
                                                               size_t width_bytes, size_t height, unsigned int value,
                                                               unsigned int esize) {
  const size_t total = width_bytes * height;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / width_bytes, col = i - row * width_bytes;
    dst[row * pitch + col] = (unsigned char)((value >> (8u * (unsigned)(col % esize))) & 0xffu);
  }
}

// fast path: one contiguous, 4-byte aligned span
__global__ __launch_bounds__(256) void graph_fill_words_kernel(unsigned int* __restrict__ dst, unsigned int pattern,
                                                               size_t n_words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = pattern;
}

// device-to-device copy of one contiguous span (16-byte vectors when both ends are aligned, bytes otherwise)
__global__ __launch_bounds__(256) void graph_copy_bytes_kernel(unsigned char* __restrict__ dst,
                                                               const unsigned char* __restrict__ src, size_t n_bytes,
                                                               int vec16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  if (vec16) {
    const size_t nv = n_bytes / 16;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) d4[i] = s4[i];
    for (size_t i = nv * 16 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_bytes; i += stride) dst[i] = src[i];
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_bytes; i += stride) dst[i] = src[i];
  }
}

}  // namespace

extern "C" hd_status hd_graph_replace_memsets(void* hip_graph, int* n_nodes_out, int* n_replaced_out) {
  if (!hip_graph) return HD_ERR_NULL;
  hipGraph_t g = reinterpret_cast<hipGraph_t>(hip_graph);
  size_t n = 0;
  if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) return HD_ERR_LAUNCH;
  std::vector<hipGraphNode_t> nodes(n);
  if (n && hipGraphGetNodes(g, nodes.data(), &n) != hipSuccess) return HD_ERR_LAUNCH;
  int replaced = 0, copies = 0;
  std::vector<char> gone(n, 0);
  for (size_t k = 0; k < n; ++k) {
    hipGraphNodeType type;
    if (hipGraphNodeGetType(nodes[k], &type) != hipSuccess) return HD_ERR_LAUNCH;
    if (type != hipGraphNodeTypeMemset) continue;
    gone[k] = 1;
    hipMemsetParams mp;
    if (hipGraphMemsetNodeGetParams(nodes[k], &mp) != hipSuccess) return HD_ERR_LAUNCH;
    size_t nd = 0, nq = 0;
    if (hipGraphNodeGetDependencies(nodes[k], nullptr, &nd) != hipSuccess) return HD_ERR_LAUNCH;
    std::vector<hipGraphNode_t> deps(nd);
    if (nd && hipGraphNodeGetDependencies(nodes[k], deps.data(), &nd) != hipSuccess) return HD_ERR_LAUNCH;
    if (hipGraphNodeGetDependentNodes(nodes[k], nullptr, &nq) != hipSuccess) return HD_ERR_LAUNCH;
    std::vector<hipGraphNode_t> outs(nq);
    if (nq && hipGraphNodeGetDependentNodes(nodes[k], outs.data(), &nq) != hipSuccess) return HD_ERR_LAUNCH;

    const size_t esize = mp.elementSize ? mp.elementSize : 1;
    const size_t width_bytes = mp.width * esize;
    const size_t height = mp.height ? mp.height : 1;
    hipKernelNodeParams kp = {};
    hipGraphNode_t knode;
    unsigned char* dst = reinterpret_cast<unsigned char*>(mp.dst);
    size_t pitch = mp.pitch ? mp.pitch : width_bytes;
    unsigned int value = mp.value;
    unsigned int es = (unsigned int)esize;
    size_t wb = width_bytes, ht = height;
    // replicate the element pattern into 32 bits
    unsigned int pattern = esize == 4 ? value
                                      : (esize == 2 ? ((value & 0xffffu) | ((value & 0xffffu) << 16))
                                                    : (value & 0xffu) * 0x01010101u);
    size_t n_words = width_bytes / 4;
    unsigned int* dst32 = reinterpret_cast<unsigned int*>(mp.dst);
    const bool fast = height == 1 && (reinterpret_cast<uintptr_t>(mp.dst) & 3u) == 0 && (width_bytes & 3u) == 0;
    void* args_fast[3] = {&dst32, &pattern, &n_words};
    void* args_gen[6] = {&dst, &pitch, &wb, &ht, &value, &es};
    const size_t items = fast ? n_words : width_bytes * height;
    if (items == 0) {  // empty memset: just drop it, keeping the ordering edges
      for (size_t a = 0; a < nd; ++a)
        for (size_t b = 0; b < nq; ++b)
          if (hipGraphAddDependencies(g, &deps[a], &outs[b], 1) != hipSuccess) return HD_ERR_LAUNCH;
      if (hipGraphDestroyNode(nodes[k]) != hipSuccess) return HD_ERR_LAUNCH;
      ++replaced;
      continue;
    }
    size_t blocks = (items + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    kp.func = fast ? reinterpret_cast<void*>(graph_fill_words_kernel) : reinterpret_cast<void*>(graph_fill_bytes_kernel);
    kp.gridDim = dim3((unsigned)blocks);
    kp.blockDim = dim3(256);
    kp.sharedMemBytes = 0;
    kp.kernelParams = fast ? args_fast : args_gen;
    kp.extra = nullptr;
    if (hipGraphAddKernelNode(&knode, g, nd ? deps.data() : nullptr, nd, &kp) != hipSuccess) return HD_ERR_LAUNCH;
    for (size_t b = 0; b < nq; ++b)
      if (hipGraphAddDependencies(g, &knode, &outs[b], 1) != hipSuccess) return HD_ERR_LAUNCH;
    if (hipGraphDestroyNode(nodes[k]) != hipSuccess) return HD_ERR_LAUNCH;
    ++replaced;
  }
  // Device-to-device memcpy nodes (contiguous same-dtype tensor copies: PyTorch's copy_ of small 1-D tensors, e.g. the
  // BatchNorm weight / bias gradients dist.GradReducer packs into its buckets) get the same treatment, as a precaution:
  // they go through the same copy-engine path as the memset nodes whose misordering was reproduced;
  // no misordered memcpy node was observed.  A one-span copy kernel with the same predecessors and successors replaces
  // each; other memcpy shapes / kinds are left alone.
  for (size_t k = 0; k < n; ++k) {
    if (gone[k]) continue;  // a memset node destroyed above
    hipGraphNodeType type;
    if (hipGraphNodeGetType(nodes[k], &type) != hipSuccess) return HD_ERR_LAUNCH;
    if (type != hipGraphNodeTypeMemcpy) continue;
    hipMemcpy3DParms cp;
    if (hipGraphMemcpyNodeGetParams(nodes[k], &cp) != hipSuccess) return HD_ERR_LAUNCH;
    if (cp.kind != hipMemcpyDeviceToDevice && cp.kind != hipMemcpyDefault) continue;
    if (cp.srcArray || cp.dstArray || cp.extent.height > 1 || cp.extent.depth > 1) continue;
    if (cp.srcPos.x || cp.srcPos.y || cp.srcPos.z || cp.dstPos.x || cp.dstPos.y || cp.dstPos.z) continue;
    if (cp.kind == hipMemcpyDefault) {  // only when both ends are device memory
      hipPointerAttribute_t a, b;
      if (hipPointerGetAttributes(&a, cp.srcPtr.ptr) != hipSuccess || hipPointerGetAttributes(&b, cp.dstPtr.ptr) != hipSuccess ||
          a.type != hipMemoryTypeDevice || b.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        continue;
      }
    }
    size_t nd = 0, nq = 0;
    if (hipGraphNodeGetDependencies(nodes[k], nullptr, &nd) != hipSuccess) return HD_ERR_LAUNCH;
    std::vector<hipGraphNode_t> deps(nd);
    if (nd && hipGraphNodeGetDependencies(nodes[k], deps.data(), &nd) != hipSuccess) return HD_ERR_LAUNCH;
    if (hipGraphNodeGetDependentNodes(nodes[k], nullptr, &nq) != hipSuccess) return HD_ERR_LAUNCH;
    std::vector<hipGraphNode_t> outs(nq);
    if (nq && hipGraphNodeGetDependentNodes(nodes[k], outs.data(), &nq) != hipSuccess) return HD_ERR_LAUNCH;
    unsigned char* dst = reinterpret_cast<unsigned char*>(cp.dstPtr.ptr);
    const unsigned char* src = reinterpret_cast<const unsigned char*>(cp.srcPtr.ptr);
    size_t bytes = cp.extent.width;
    if (bytes == 0) {
      for (size_t a = 0; a < nd; ++a)
        for (size_t b = 0; b < nq; ++b)
          if (hipGraphAddDependencies(g, &deps[a], &outs[b], 1) != hipSuccess) return HD_ERR_LAUNCH;
      if (hipGraphDestroyNode(nodes[k]) != hipSuccess) return HD_ERR_LAUNCH;
      ++copies;
      continue;
    }
    int vec16 = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0 ? 1 : 0;
    void* args[4] = {&dst, &src, &bytes, &vec16};
    size_t items = vec16 ? (bytes + 15) / 16 : bytes;
    size_t blocks = (items + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipKernelNodeParams kp = {};
    kp.func = reinterpret_cast<void*>(graph_copy_bytes_kernel);
    kp.gridDim = dim3((unsigned)blocks);
    kp.blockDim = dim3(256);
    kp.sharedMemBytes = 0;
    kp.kernelParams = args;
    kp.extra = nullptr;
    hipGraphNode_t knode;
    if (hipGraphAddKernelNode(&knode, g, nd ? deps.data() : nullptr, nd, &kp) != hipSuccess) return HD_ERR_LAUNCH;
    for (size_t b = 0; b < nq; ++b)
      if (hipGraphAddDependencies(g, &knode, &outs[b], 1) != hipSuccess) return HD_ERR_LAUNCH;
    if (hipGraphDestroyNode(nodes[k]) != hipSuccess) return HD_ERR_LAUNCH;
    ++copies;
  }
  if (n_nodes_out) *n_nodes_out = (int)n;
  if (n_replaced_out) *n_replaced_out = replaced + copies;
  return HD_OK;
}

// ---- splitting a captured step at bucket boundaries (multi-GPU overlap under hipGraph replay) -------------------
// A train step replayed as ONE graph cannot overlap its gradient all-reduce with backward: a collective can only be
// launched once the whole replay has been enqueued.  So the capture carries MARKERS — a one-lane kernel launched by
// dist.GradReducer right after it packed a bucket — and hd_graph_split_instantiate() cuts the captured graph into
// segments ending at the markers: segment k = every node that marker k depends on and that no earlier marker does,
// the tail = the rest.  Each segment is a clone of the graph with the other nodes destroyed (the clone keeps kernel
// parameters and the intra-segment edges; edges across segments disappear with the destroyed endpoints and are
// re-established by launching the segments in order on one stream).  The host then replays segment k, launches
// bucket k's RCCL all-reduce on RCCL's stream, replays segment k + 1 meanwhile, ... (train.Trainer._replay_split).
namespace {

__global__ void graph_mark_kernel(int32_t* __restrict__ slot, int32_t index) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *slot = index;
}

struct GraphExecs {
  std::vector<hipGraphExec_t> execs;
  std::vector<hipGraph_t> graphs;
};

}  // namespace

extern "C" hd_status hd_graph_mark(int32_t* slot, int index, hd_stream stream) {
  if (!slot) return HD_ERR_NULL;
  hipLaunchKernelGGL(graph_mark_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slot, (int32_t)index);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_graph_split_instantiate(void* hip_graph, const int32_t* slot, int max_segments, void** out_execs,
                                                int* marker_index_out, int* n_segments_out, int* n_nodes_out) {
  if (!hip_graph || !slot || !out_execs || !marker_index_out || !n_segments_out) return HD_ERR_NULL;
  if (max_segments < 1) return HD_ERR_SHAPE;
  hipGraph_t g = reinterpret_cast<hipGraph_t>(hip_graph);
  size_t n = 0;
  if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) return HD_ERR_LAUNCH;
  std::vector<hipGraphNode_t> nodes(n);
  if (n && hipGraphGetNodes(g, nodes.data(), &n) != hipSuccess) return HD_ERR_LAUNCH;
  // predecessor lists
  std::vector<std::vector<int>> pred(n);
  auto index_of = [&](hipGraphNode_t x) -> int {
    for (size_t i = 0; i < n; ++i)
      if (nodes[i] == x) return (int)i;
    return -1;
  };
  for (size_t i = 0; i < n; ++i) {
    size_t nd = 0;
    if (hipGraphNodeGetDependencies(nodes[i], nullptr, &nd) != hipSuccess) return HD_ERR_LAUNCH;
    std::vector<hipGraphNode_t> deps(nd);
    if (nd && hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd) != hipSuccess) return HD_ERR_LAUNCH;
    for (size_t d = 0; d < nd; ++d) {
      const int j = index_of(deps[d]);
      if (j < 0) return HD_ERR_LAUNCH;
      pred[i].push_back(j);
    }
  }
  // markers: kernel nodes of graph_mark_kernel writing to `slot`
  struct Marker { int node, index; };
  std::vector<Marker> markers;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType type;
    if (hipGraphNodeGetType(nodes[i], &type) != hipSuccess) return HD_ERR_LAUNCH;
    if (type != hipGraphNodeTypeKernel) continue;
    hipKernelNodeParams kp;
    if (hipGraphKernelNodeGetParams(nodes[i], &kp) != hipSuccess) return HD_ERR_LAUNCH;
    if (kp.func != reinterpret_cast<void*>(graph_mark_kernel) || !kp.kernelParams) continue;
    if (*reinterpret_cast<int32_t**>(kp.kernelParams[0]) != slot) continue;
    markers.push_back({(int)i, (int)*reinterpret_cast<int32_t*>(kp.kernelParams[1])});
  }
  const int K = (int)markers.size();
  if (K + 1 > max_segments) return HD_ERR_SHAPE;
  // ancestors of every marker; markers are ordered by ancestry (captured from one stream: a chain)
  std::vector<std::vector<char>> anc(K, std::vector<char>(n, 0));
  for (int k = 0; k < K; ++k) {
    std::vector<int> stack{markers[k].node};
    anc[k][markers[k].node] = 1;
    while (!stack.empty()) {
      const int u = stack.back();
      stack.pop_back();
      for (int p : pred[u])
        if (!anc[k][p]) {
          anc[k][p] = 1;
          stack.push_back(p);
        }
    }
  }
  std::vector<int> order(K);
  for (int k = 0; k < K; ++k) order[k] = k;
  std::sort(order.begin(), order.end(), [&](int a, int b) {  // a before b iff a is an ancestor of b
    if (anc[b][markers[a].node]) return true;
    if (anc[a][markers[b].node]) return false;
    return markers[a].index < markers[b].index;
  });
  std::vector<int> seg(n, K);  // K = tail
  for (int r = K - 1; r >= 0; --r)  // later markers first, earlier ones overwrite: seg = first marker that needs the node
    for (size_t i = 0; i < n; ++i)
      if (anc[order[r]][i]) seg[i] = r;
  GraphExecs made;
  int produced = 0;
  for (int s = 0; s <= K; ++s) {
    size_t count = 0;
    for (size_t i = 0; i < n; ++i) count += seg[i] == s;
    if (count == 0) {
      if (s < K) return HD_ERR_LAUNCH;  // a marker segment holds at least its marker
      break;                            // empty tail
    }
    hipGraph_t c = nullptr;
    if (hipGraphClone(&c, g) != hipSuccess) return HD_ERR_LAUNCH;
    made.graphs.push_back(c);
    for (size_t i = 0; i < n; ++i) {
      if (seg[i] == s) continue;
      hipGraphNode_t cn = nullptr;
      if (hipGraphNodeFindInClone(&cn, nodes[i], c) != hipSuccess) return HD_ERR_LAUNCH;
      if (hipGraphDestroyNode(cn) != hipSuccess) return HD_ERR_LAUNCH;
    }
    hipGraphExec_t ex = nullptr;
    if (hipGraphInstantiate(&ex, c, nullptr, nullptr, 0) != hipSuccess) return HD_ERR_LAUNCH;
    made.execs.push_back(ex);
    out_execs[produced] = reinterpret_cast<void*>(ex);
    if (s < K) marker_index_out[produced] = markers[order[s]].index;
    ++produced;
  }
  *n_segments_out = produced;
  if (n_nodes_out) *n_nodes_out = (int)n;
  // the pruned clones must outlive their executables only until instantiation on this stack
  for (hipGraph_t c : made.graphs) (void)hipGraphDestroy(c);
  return HD_OK;
}

extern "C" hd_status hd_graph_exec_launch(void* exec, hd_stream stream) {
  if (!exec) return HD_ERR_NULL;
  return hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(exec), (hipStream_t)stream) == hipSuccess ? HD_OK : HD_ERR_LAUNCH;
}

extern "C" hd_status hd_graph_exec_destroy(void* exec) {
  if (!exec) return HD_OK;
  return hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(exec)) == hipSuccess ? HD_OK : HD_ERR_LAUNCH;
}
