// multi_stream_host.cpp -- C++ front end for BASELINE.json configs[3]: independent camera streams sharded over the GPUs
// of one node, one detector handle and one host thread per GPU, and exactly one collective: an RCCL broadcast (over
// xGMI) of the per-stream parameter block -- intrinsics, tag size, decimation -- from device 0 at start-up.
//
// Frames are independent units (the reference handles one frame at a time with no cross-frame state,
// isaac_ros_apriltag/src/apriltag_node.cpp:613-623), so stream s is owned by GPU s % G and the data path needs no
// collective; results return by per-GPU D2H inside amdAprilTagsDetectBatch.  Single process, one communicator over the
// local devices (ncclCommInitAll), no MPI -- SURVEY.md section 8(e).
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude examples/multi_stream_host.cpp \
//         -Lisaac_ros_apriltag_amd -lapriltag_amd -lrccl -Wl,-rpath,$PWD/isaac_ros_apriltag_amd -o examples/multi_stream_host
//   python tools/dump_streams.py streams.bin            # frames + per-stream parameters (same generator as bench.py)
//   ./examples/multi_stream_host streams.bin [gpus] [steps] [--host-frames] [--shared-gpu]
//
// --shared-gpu: the G "GPUs" are G ranks on device 0 -- G worker threads, G handles (each with its own streams, side streams and
// pinned blocks), G communicator ranks -- so that a one-GPU box exercises what the 8-GPU run relies on inside the library: the
// family registry's lock, the device guard of every entry point, eight handles' launch sequences side by side on one device.
// RCCL refuses a communicator with one device twice; the flag then (and only then) replaces the broadcast by a device-to-device
// fan-out of the same block from rank 0's copy, and the JSON line says which of the two ran.
//
// --host-frames: the frames start in (pinned) HOST memory every step, as they do behind a camera driver.  Every GPU's thread
// double-buffers: amdAprilTagsSubmitBatch on buffer A returns at once, the next step's frames are copied into buffer B on a
// copy stream of the thread's own while the detector runs, amdAprilTagsWaitBatch collects; "fps_host_frames" is that rate
// (PCIe-inclusive: 8 x 1080p streams at 14 000 frames/s per GPU are 29 GB/s of input per GPU).
//
// Input file: int32 magic 'ATS1', streams S, frames-per-stream F, width, height, decimate; then S records of five
// doubles {fx, fy, cx, cy, tag_size}; then S*F mono8 frames.  Output: one JSON line with the whole-job frame rate and,
// per stream, the detection count and an FNV-1a checksum of (id, corners, translation) that tests/test_gpu_parity.py
// compares with the Python path.  The tag size is a property of the handle (amdCreateAprilTagsDetector, as in
// nvCreateAprilTagsDetector): a GPU whose streams carry different tag sizes gets one handle per distinct size and runs
// them one after the other.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "apriltag_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(3); } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); exit(4); } } while (0)
#define CHECK_AT(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s: error code %d\n", #x, r_); exit(5); } } while (0)

struct StreamParams { double fx, fy, cx, cy, tag_size; };

struct Barrier {   // reusable spin barrier for the few host threads
  std::atomic<int> count{0}, gen{0};
  int n;
  explicit Barrier(int n_) : n(n_) {}
  void wait() {
    const int g = gen.load();
    if (count.fetch_add(1) + 1 == n) { count.store(0); gen.fetch_add(1); }
    else while (gen.load() == g) std::this_thread::yield();
  }
};

static uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
  const unsigned char* b = static_cast<const unsigned char*>(p);
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001B3ull; }
  return h;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s streams.bin [gpus] [steps] [--host-frames] [--shared-gpu]\n", argv[0]); return 2; }
  bool host_frames = false, shared_gpu = false;
  for (int i = 2; i < argc; i++) {
    const bool hf = !strcmp(argv[i], "--host-frames"), sg = !strcmp(argv[i], "--shared-gpu");
    if (hf || sg) { host_frames |= hf; shared_gpu |= sg; for (int j = i; j + 1 < argc; j++) argv[j] = argv[j + 1]; argc--; i--; }
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  int32_t hdr[6];
  if (fread(hdr, 4, 6, f) != 6 || hdr[0] != 0x31535441) { fprintf(stderr, "bad header\n"); return 2; }
  const int S = hdr[1], F = hdr[2], W = hdr[3], H = hdr[4], decimate = hdr[5];
  std::vector<StreamParams> params(S);
  if (fread(params.data(), sizeof(StreamParams), S, f) != (size_t)S) { fprintf(stderr, "short file\n"); return 2; }
  const size_t fbytes = (size_t)W * H;
  std::vector<uint8_t> frames((size_t)S * F * fbytes);
  if (fread(frames.data(), 1, frames.size(), f) != frames.size()) { fprintf(stderr, "short file\n"); return 2; }
  fclose(f);

  int ndev = 0;
  CHECK_HIP(hipGetDeviceCount(&ndev));
  int G = argc > 2 ? atoi(argv[2]) : ndev;
  if (G < 1 || (G > ndev && !shared_gpu) || G > 64) { fprintf(stderr, "%d GPUs requested, %d visible\n", G, ndev); return 2; }
  auto dev_of = [&](int rank) { return shared_gpu ? 0 : rank; };
  const int steps = argc > 3 ? atoi(argv[3]) : 5;

  // ---- the one collective: broadcast of the parameter block from device 0 (RCCL over xGMI) ---------------------
  std::vector<int> devs(G);
  for (int i = 0; i < G; i++) devs[i] = dev_of(i);
  std::vector<ncclComm_t> comms(G, nullptr);
  bool rccl = true;
  if (shared_gpu && G > 1) {   // RCCL may refuse one device twice: only then, and only behind the flag, the fan-out below
    rccl = ncclCommInitAll(comms.data(), G, devs.data()) == ncclSuccess;
    if (!rccl) for (auto& c : comms) c = nullptr;
  } else {
    CHECK_NCCL(ncclCommInitAll(comms.data(), G, devs.data()));
  }
  std::vector<hipStream_t> streams(G);
  std::vector<double*> d_block(G);
  const size_t block_doubles = (size_t)S * 5 + 1;   // parameters + decimate
  for (int i = 0; i < G; i++) {
    CHECK_HIP(hipSetDevice(dev_of(i)));
    CHECK_HIP(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking));
    CHECK_HIP(hipMalloc((void**)&d_block[i], block_doubles * 8));
    CHECK_HIP(hipMemset(d_block[i], 0, block_doubles * 8));
  }
  {
    std::vector<double> host(block_doubles);
    memcpy(host.data(), params.data(), (size_t)S * 5 * 8);
    host[(size_t)S * 5] = (double)decimate;
    CHECK_HIP(hipSetDevice(0));
    CHECK_HIP(hipMemcpy(d_block[0], host.data(), block_doubles * 8, hipMemcpyHostToDevice));   // only device 0 holds it
  }
  for (int i = 0; i < G; i++) { CHECK_HIP(hipSetDevice(dev_of(i))); CHECK_HIP(hipDeviceSynchronize()); }   // the fills above, ahead of the non-blocking streams
  if (rccl) {
    CHECK_NCCL(ncclGroupStart());
    for (int i = 0; i < G; i++) {
      CHECK_HIP(hipSetDevice(dev_of(i)));
      CHECK_NCCL(ncclBroadcast(d_block[i], d_block[i], block_doubles, ncclDouble, 0, comms[i], streams[i]));
    }
    CHECK_NCCL(ncclGroupEnd());
  } else {   // (--shared-gpu only) the same block to every rank's copy, device to device on the rank's stream
    for (int i = 1; i < G; i++) CHECK_HIP(hipMemcpyAsync(d_block[i], d_block[0], block_doubles * 8, hipMemcpyDeviceToDevice, streams[i]));
  }
  for (int i = 0; i < G; i++) { CHECK_HIP(hipSetDevice(dev_of(i))); CHECK_HIP(hipStreamSynchronize(streams[i])); }

  // ---- one host thread, one handle, one share of the streams per GPU ------------------------------------------
  Barrier bar(G);
  std::vector<double> seconds(G, 0.0), seconds_host(G, 0.0);
  std::vector<uint64_t> sums(S, 0);
  std::vector<uint32_t> ndet(S, 0);
  const uint32_t max_tags = 64;
  auto worker = [&](int g) {
    CHECK_HIP(hipSetDevice(dev_of(g)));
    // this GPU's copy of the block, as received through the broadcast
    std::vector<double> blk(block_doubles);
    CHECK_HIP(hipMemcpyAsync(blk.data(), d_block[g], block_doubles * 8, hipMemcpyDeviceToHost, streams[g]));
    CHECK_HIP(hipStreamSynchronize(streams[g]));
    std::vector<int> mine;
    for (int s = 0; s < S; s++) if (s % G == g) mine.push_back(s);
    if (mine.empty()) { bar.wait(); bar.wait(); return; }
    // one handle per distinct tag size among this GPU's streams
    std::map<float, std::vector<int>> by_size;
    for (int s : mine) by_size[(float)blk[(size_t)s * 5 + 4]].push_back(s);
    struct Group {
      std::vector<int> streams;
      amdAprilTagsHandle h = nullptr;
      uint8_t* d_frames = nullptr;
      uint8_t* d_frames_b = nullptr;      // --host-frames: the second device buffer
      uint8_t* h_frames = nullptr;        //                and the group's frames in pinned host memory
      std::vector<amdAprilTagsImageInput_t> imgs_b;
      std::vector<amdAprilTagsImageInput_t> imgs;
      std::vector<amdAprilTagsCameraIntrinsics_t> intr;
      std::vector<amdAprilTagsID_t> tags;
      std::vector<uint32_t> cnt;
    };
    std::vector<Group> groups;
    for (auto& kv : by_size) {
      Group gr;
      gr.streams = kv.second;
      const uint32_t B = (uint32_t)(gr.streams.size() * F);
      amdAprilTagsConfig_t cfg;
      amdAprilTagsDefaultConfig(&cfg, (uint32_t)W, (uint32_t)H);
      cfg.decimate = (uint32_t)blk[(size_t)S * 5];
      cfg.max_batch = B;
      cfg.device = dev_of(g);
      cfg.tag_size = kv.first;
      CHECK_AT(amdCreateAprilTagsDetectorEx(&gr.h, &cfg));
      CHECK_HIP(hipMalloc((void**)&gr.d_frames, (size_t)B * fbytes));
      gr.imgs.resize(B); gr.intr.resize(B); gr.tags.resize((size_t)B * max_tags); gr.cnt.resize(B);
      for (size_t k = 0; k < gr.streams.size(); k++) {
        const int s = gr.streams[k];
        // (on the rank's own stream, not the legacy stream: with several ranks on one device -- --shared-gpu -- a legacy-stream copy
        // of this thread would meet the launch-graph capture of another thread's handle and fail)
        CHECK_HIP(hipMemcpyAsync(gr.d_frames + k * F * fbytes, frames.data() + (size_t)s * F * fbytes, (size_t)F * fbytes, hipMemcpyHostToDevice, streams[g]));
        CHECK_HIP(hipStreamSynchronize(streams[g]));
        for (int i = 0; i < F; i++) {
          const size_t b = k * F + i;
          gr.imgs[b] = {(uint32_t)W, (uint32_t)H, gr.d_frames + b * fbytes, (size_t)W};
          gr.intr[b] = {(float)blk[(size_t)s * 5 + 0], (float)blk[(size_t)s * 5 + 1], (float)blk[(size_t)s * 5 + 2], (float)blk[(size_t)s * 5 + 3]};
        }
      }
      if (host_frames) {
        CHECK_HIP(hipMalloc((void**)&gr.d_frames_b, (size_t)B * fbytes));
        CHECK_HIP(hipHostMalloc((void**)&gr.h_frames, (size_t)B * fbytes));
        gr.imgs_b = gr.imgs;
        for (size_t k = 0; k < gr.streams.size(); k++)
          memcpy(gr.h_frames + k * F * fbytes, frames.data() + (size_t)gr.streams[k] * F * fbytes, (size_t)F * fbytes);
        for (size_t b = 0; b < gr.imgs_b.size(); b++) gr.imgs_b[b].dev_ptr = gr.d_frames_b + b * fbytes;
      }
      groups.push_back(std::move(gr));
    }
    auto run_all = [&]() {
      for (Group& gr : groups)
        CHECK_AT(amdAprilTagsDetectBatch(gr.h, (uint32_t)gr.imgs.size(), gr.imgs.data(), gr.intr.data(), gr.tags.data(), gr.cnt.data(), max_tags, nullptr));
    };
    run_all();   // warm-up
    bar.wait();
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < steps; it++) run_all();
    seconds[g] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (host_frames) {
      // double-buffered, PCIe-inclusive: submit buffer `cur`, copy the next step's frames into the other buffer meanwhile, wait
      hipStream_t copy_stream;
      CHECK_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
      for (Group& gr : groups) CHECK_HIP(hipMemcpyAsync(gr.d_frames, gr.h_frames, gr.imgs.size() * fbytes, hipMemcpyHostToDevice, copy_stream));
      CHECK_HIP(hipStreamSynchronize(copy_stream));
      const auto th0 = std::chrono::steady_clock::now();
      for (int it = 0; it < steps; it++) {
        for (Group& gr : groups) {
          const bool odd = it & 1;
          CHECK_AT(amdAprilTagsSubmitBatch(gr.h, (uint32_t)gr.imgs.size(), odd ? gr.imgs_b.data() : gr.imgs.data(), gr.intr.data(), max_tags, nullptr));
          CHECK_HIP(hipMemcpyAsync(odd ? gr.d_frames : gr.d_frames_b, gr.h_frames, gr.imgs.size() * fbytes, hipMemcpyHostToDevice, copy_stream));
          CHECK_AT(amdAprilTagsWaitBatch(gr.h, gr.tags.data(), gr.cnt.data()));
          CHECK_HIP(hipStreamSynchronize(copy_stream));
        }
      }
      seconds_host[g] = std::chrono::duration<double>(std::chrono::steady_clock::now() - th0).count();
      CHECK_HIP(hipStreamDestroy(copy_stream));
    }
    bar.wait();
    for (Group& gr : groups) {
      for (size_t k = 0; k < gr.streams.size(); k++) {
        uint64_t hsh = 0xCBF29CE484222325ull;
        uint32_t n = 0;
        for (int i = 0; i < F; i++) {
          const size_t b = k * F + i;
          for (uint32_t d = 0; d < gr.cnt[b]; d++) {
            const amdAprilTagsID_t& t = gr.tags[b * max_tags + d];
            hsh = fnv1a(hsh, &t.id, sizeof(t.id));
            hsh = fnv1a(hsh, t.corners, sizeof(t.corners));
            hsh = fnv1a(hsh, t.translation, sizeof(t.translation));
          }
          n += gr.cnt[b];
        }
        sums[gr.streams[k]] = hsh;
        ndet[gr.streams[k]] = n;
      }
      CHECK_HIP(hipFree(gr.d_frames));
      if (gr.d_frames_b) CHECK_HIP(hipFree(gr.d_frames_b));
      if (gr.h_frames) CHECK_HIP(hipHostFree(gr.h_frames));
      CHECK_AT(amdAprilTagsDestroy(gr.h));
    }
  };
  std::vector<std::thread> th;
  for (int g = 0; g < G; g++) th.emplace_back(worker, g);
  for (auto& t : th) t.join();
  double worst = 0;
  for (double s : seconds) worst = s > worst ? s : worst;
  for (int i = 0; i < G; i++) { CHECK_HIP(hipSetDevice(dev_of(i))); (void)hipFree(d_block[i]); (void)hipStreamDestroy(streams[i]); if (comms[i]) ncclCommDestroy(comms[i]); }

  printf("{\"gpus\": %d, \"shared_gpu\": %s, \"streams\": %d, \"frames_per_stream\": %d, \"steps\": %d, \"fps\": %.1f, \"collective\": \"%s of %zu doubles\", ",
         G, shared_gpu ? "true" : "false", S, F, steps, worst > 0 ? (double)S * F * steps / worst : 0.0,
         rccl ? "ncclBroadcast" : "device-to-device fan-out (RCCL refused one device twice)", block_doubles);
  // wall time of every GPU's own timed loop (a straggler shows here; fps uses the slowest)
  if (host_frames) {
    double worst_h = 0;
    for (double sh : seconds_host) worst_h = sh > worst_h ? sh : worst_h;
    printf("\"fps_host_frames\": %.1f, ", worst_h > 0 ? (double)S * F * steps / worst_h : 0.0);
  }
  printf("\"per_gpu_seconds\": [");
  for (int g = 0; g < G; g++) printf("%s%.6f", g ? ", " : "", seconds[g]);
  printf("], \"streams_out\": [");
  for (int s = 0; s < S; s++) printf("%s{\"stream\": %d, \"gpu\": %d, \"detections\": %u, \"fnv\": \"%016llx\"}", s ? ", " : "", s, s % G, ndet[s], (unsigned long long)sums[s]);
  printf("]}\n");
  return 0;
}
