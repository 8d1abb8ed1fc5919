// repro_graph_regrow.hip -- standalone reproducer, NO library code: the HIP call pattern of a detector handle whose buffers are
// reallocated between two stream captures ("host crash in the capacity-regrowth path").
//
//   per "handle": one non-blocking stream + three side streams, fork / join events, a pinned host block the first kernel reads
//   and the last kernel stamps; capture -> instantiate -> launch -> wait; free + allocate the buffers again (null-stream fills,
//   device-wide wait, as the library's regrowth does); capture -> instantiate -> launch -> wait again; destroy everything.
//
// Build:  hipcc --offload-arch=gfx950 -O2 tools/repro_graph_regrow.hip -o tools/repro_graph_regrow
// Run:    tools/repro_graph_regrow [handles=300] [destroy|retire|keep] [branches=3] [chain=0]
//   chain  : extra kernel nodes ahead of the fork, each with a 1 280-byte by-value argument and 70 KB of dynamic LDS
//   destroy: hipGraphExecDestroy of the first graph right before the second capture (what the library did until round 4)
//   retire : the first graph is destroyed with the handle (what round 5 ships)
//   keep   : no regrowth at all (control)
//   destroy_nosync: as destroy, but the "regrowth" is the library's candidate-list growth inside a call -- one hipMalloc + hipFree,
//            no fill, NO device-wide wait between the first graph's stream wait and its destruction -- with the fork event
//            recorded twice and one side stream left without a kernel, as the library's captured sequence has them
// Prints the number of handles that completed and of launches whose stream wait returned before the stamp was visible.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(e)                                                                                          \
  do {                                                                                                    \
    hipError_t _e = (e);                                                                                  \
    if (_e != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #e, hipGetErrorString(_e), __LINE__); exit(2); } \
  } while (0)

struct Block { uint32_t seq; uint32_t n; uint32_t stamp; uint32_t sum; };
struct BigArgs { uint32_t w[320]; };   // 1 280 bytes by value: the library passes its parameter struct (about 1.1 KB) to every kernel

__global__ void k_first(const Block* host, Block* dev, uint32_t* a, uint32_t n) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *dev = *host;   // reads the pinned block over the bus, like the library's k_prologue
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) a[i] = i;
}
__global__ void k_branch(const uint32_t* a, uint32_t* b, uint32_t n, uint32_t mul) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) b[i] = a[i] * mul + 1u;
}
__global__ void k_chain(uint32_t* a, uint32_t n, BigArgs big) {   // (a chain of these stands for the library's ~20 launches)
  extern __shared__ uint32_t dyn[];
  if (threadIdx.x < 64) dyn[threadIdx.x] = big.w[threadIdx.x];
  __syncthreads();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) a[i] += dyn[i & 63] & 1u;
}
__global__ void k_last(const Block* dev, Block* host, const uint32_t* b0, const uint32_t* b1, const uint32_t* b2, uint32_t n) {
  __shared__ uint32_t s;
  if (threadIdx.x == 0) s = 0;
  __syncthreads();
  uint32_t acc = 0;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) acc += b0[i] ^ b1[i] ^ b2[i];
  atomicAdd(&s, acc);
  __syncthreads();
  if (threadIdx.x == 0) { host->sum = s; host->n = n; __threadfence_system(); host->stamp = dev->seq; }
}

struct Handle {
  hipStream_t s = nullptr, aux[8] = {};
  hipEvent_t fork = nullptr, join[8] = {};
  uint32_t *a = nullptr, *b[8] = {};
  Block *h = nullptr, *d = nullptr;
  uint32_t n = 0;
  std::vector<hipGraphExec_t> retired;
};

static void alloc_buffers(Handle& H, uint32_t n, int nb) {
  if (H.a) { CHECK(hipFree(H.a)); for (int i = 0; i < nb; i++) CHECK(hipFree(H.b[i])); }
  H.n = n;
  CHECK(hipMalloc((void**)&H.a, (size_t)n * 4));
  for (int i = 0; i < nb; i++) CHECK(hipMalloc((void**)&H.b[i], (size_t)n * 4));
  CHECK(hipMemset(H.a, 0xFF, (size_t)n * 4));   // null-stream fills + a device-wide wait: the library's clear_hash_tables
  CHECK(hipMemset(H.b[0], 0, (size_t)n * 4));
  CHECK(hipDeviceSynchronize());
}

static int g_chain = 0;   // extra nodes with big by-value arguments and dynamic LDS ahead of the fork
static bool g_libshape = false;   // destroy_nosync: fork event recorded twice, last side stream without a kernel
static void enqueue(Handle& H, int nb) {
  hipLaunchKernelGGL(k_first, dim3(64), dim3(256), 0, H.s, H.h, H.d, H.a, H.n);
  if (g_chain) {
    BigArgs big;
    for (int i = 0; i < 320; i++) big.w[i] = 0;
    for (int k = 0; k < g_chain; k++) hipLaunchKernelGGL(k_chain, dim3(64), dim3(256), 70000, H.s, H.a, H.n, big);
  }
  CHECK(hipEventRecord(H.fork, H.s));
  if (g_libshape) {
    hipLaunchKernelGGL(k_branch, dim3(64), dim3(256), 0, H.s, H.a, H.b[nb - 1], H.n, 1u);
    CHECK(hipEventRecord(H.fork, H.s));
  }
  for (int i = 0; i < nb; i++) CHECK(hipStreamWaitEvent(H.aux[i], H.fork, 0));
  for (int i = 0; i < nb - (g_libshape ? 1 : 0); i++) hipLaunchKernelGGL(k_branch, dim3(64), dim3(256), 0, H.aux[i], H.a, H.b[i], H.n, (uint32_t)(3 + 2 * i));
  for (int i = 0; i < nb; i++) { CHECK(hipEventRecord(H.join[i], H.aux[i])); CHECK(hipStreamWaitEvent(H.s, H.join[i], 0)); }
  hipLaunchKernelGGL(k_last, dim3(1), dim3(256), 0, H.s, H.d, H.h, H.b[0], H.b[1 % nb], H.b[2 % nb], H.n);
}

static hipGraphExec_t capture(Handle& H, int nb) {
  hipGraph_t g = nullptr;
  hipGraphExec_t e = nullptr;
  CHECK(hipStreamBeginCapture(H.s, hipStreamCaptureModeThreadLocal));
  enqueue(H, nb);
  CHECK(hipStreamEndCapture(H.s, &g));
  CHECK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
  CHECK(hipGraphDestroy(g));
  return e;
}

int main(int argc, char** argv) {
  const int handles = argc > 1 ? atoi(argv[1]) : 300;
  const char* mode = argc > 2 ? argv[2] : "destroy";
  const int nb = argc > 3 ? atoi(argv[3]) : 3;
  g_chain = argc > 4 ? atoi(argv[4]) : 0;
  g_libshape = !strcmp(mode, "destroy_nosync");
  if (g_chain) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain), hipFuncAttributeMaxDynamicSharedMemorySize, 157000));
  uint32_t seq = 0, late = 0, wrong = 0;
  for (int it = 0; it < handles; it++) {
    Handle H;
    CHECK(hipStreamCreateWithFlags(&H.s, hipStreamNonBlocking));
    for (int i = 0; i < nb; i++) CHECK(hipStreamCreateWithFlags(&H.aux[i], hipStreamNonBlocking));
    CHECK(hipEventCreateWithFlags(&H.fork, hipEventDisableTiming));
    for (int i = 0; i < nb; i++) CHECK(hipEventCreateWithFlags(&H.join[i], hipEventDisableTiming));
    CHECK(hipHostMalloc((void**)&H.h, sizeof(Block)));
    CHECK(hipMalloc((void**)&H.d, sizeof(Block)));
    memset(H.h, 0, sizeof(Block));
    alloc_buffers(H, 1u << 18, nb);
    hipGraphExec_t e1 = capture(H, nb), e2 = nullptr;
    auto run = [&](hipGraphExec_t e) {
      H.h->seq = ++seq;
      CHECK(hipGraphLaunch(e, H.s));
      CHECK(hipStreamSynchronize(H.s));
      if (((volatile Block*)H.h)->stamp != seq) { late++; CHECK(hipDeviceSynchronize()); if (((volatile Block*)H.h)->stamp != seq) wrong++; }
    };
    run(e1); run(e1);
    if (!strcmp(mode, "destroy_nosync")) {
      for (int rep = 0; rep < 2; rep++) {   // (the stress loop's handle regrows twice)
        uint32_t* nb0 = nullptr;
        CHECK(hipMalloc((void**)&nb0, (size_t)H.n * 4));
        CHECK(hipFree(H.b[0]));
        H.b[0] = nb0;
        CHECK(hipGraphExecDestroy(e2 ? e2 : e1));
        e2 = capture(H, nb);
        run(e2);
      }
    } else if (strcmp(mode, "keep")) {
      alloc_buffers(H, 1u << 19, nb);   // "regrowth": the captured launches carry the old pointers
      if (!strcmp(mode, "destroy")) CHECK(hipGraphExecDestroy(e1)); else H.retired.push_back(e1);
      e2 = capture(H, nb);
      run(e2); run(e2);
    } else {
      H.retired.push_back(e1);
    }
    CHECK(hipDeviceSynchronize());
    if (e2) CHECK(hipGraphExecDestroy(e2));
    for (hipGraphExec_t e : H.retired) CHECK(hipGraphExecDestroy(e));
    CHECK(hipFree(H.a)); for (int i = 0; i < nb; i++) CHECK(hipFree(H.b[i]));
    CHECK(hipFree(H.d)); CHECK(hipHostFree(H.h));
    CHECK(hipStreamDestroy(H.s)); for (int i = 0; i < nb; i++) CHECK(hipStreamDestroy(H.aux[i]));
    CHECK(hipEventDestroy(H.fork)); for (int i = 0; i < nb; i++) CHECK(hipEventDestroy(H.join[i]));
  }
  printf("repro_graph_regrow: %d handles (%s, %d branches, chain %d) completed, late stream waits %u, stamps still missing after a device wait %u\n",
         handles, mode, nb, g_chain, late, wrong);
  return 0;
}
