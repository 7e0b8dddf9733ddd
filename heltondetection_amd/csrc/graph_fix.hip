// graph_fix.hip — rewrite the memset nodes of a captured hipGraph into kernel nodes.
//
// Why: on ROCm 7.2 / MI355X a replayed graph that contains memset nodes (hipMemsetAsync captured
// from this library's callers — in practice MIOpen's conv-backward solvers, which zero their
// split-K accumulators that way) was observed to corrupt results: the memset is not reliably
// ordered against its neighbouring kernel nodes.  Kernel nodes are.  This entry
// point takes the raw hipGraph_t BEFORE instantiation (PyTorch: CUDAGraph(keep_graph=True) ->
// raw_cuda_graph()), and for every memset node adds a fill-kernel node with the same
// predecessors and successors, then destroys the memset node.  The caller instantiates afterwards.
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

}  // namespace

extern "C" hd_status hd_graph_replace_memsets(void* hip_graph, int* n_nodes_out, int* n_replaced_out) {
  if (!hip_graph) return HD_ERR_NULL;
  hipGraph_t g = reinterpret_cast<hipGraph_t>(hip_graph);
  size_t n = 0;
  if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) return HD_ERR_LAUNCH;
  std::vector<hipGraphNode_t> nodes(n);
  if (n && hipGraphGetNodes(g, nodes.data(), &n) != hipSuccess) return HD_ERR_LAUNCH;
  int replaced = 0;
  for (size_t k = 0; k < n; ++k) {
    hipGraphNodeType type;
    if (hipGraphNodeGetType(nodes[k], &type) != hipSuccess) return HD_ERR_LAUNCH;
    if (type != hipGraphNodeTypeMemset) continue;
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
  if (n_nodes_out) *n_nodes_out = (int)n;
  if (n_replaced_out) *n_replaced_out = replaced;
  return HD_OK;
}
