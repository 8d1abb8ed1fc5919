// Drives the ROS-free node shell (isaac_ros_apriltag_amd/csrc/node_shell.cpp, compiled into this program) under
// -fsanitize=address,undefined through its C harness: constructor validation with the reference's error text, stamp
// mismatch, bad encoding, bad step, and -- on a machine without a HIP device -- the failed detector creation that the
// shell reports and retries.  The detector library is dlopen'ed by the shell as usual (not instrumented).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

struct NodeShellHarness;
struct NodeShellDetection;
extern "C" {
NodeShellHarness* node_shell_create(int max_tags, double size, int tile_size, const char* tag_family, const char* backends,
                                    int decimate, char* err, size_t err_len);
void node_shell_destroy(NodeShellHarness* h);
int node_shell_on_frame(NodeShellHarness* h, const uint8_t* data, int is_device, const char* encoding, uint32_t width,
                        uint32_t height, uint32_t step, const double* k9, const char* frame_id, int32_t sec, uint32_t nanosec,
                        int32_t info_sec, uint32_t info_nanosec, NodeShellDetection* out, int max_out, char* out_frame_id,
                        size_t frame_id_len, char* err, size_t err_len);
}

int main() {
  char err[512];
  err[0] = 0;
  if (node_shell_create(64, 0.22, 4, "NOTHING", "CUDA", 1, err, sizeof(err)) != nullptr) return 2;
  if (!strstr(err, "Tag family not supported by specified backend")) { fprintf(stderr, "unexpected: %s\n", err); return 3; }
  if (node_shell_create(64, 0.22, 4, "tag25h9", "CUDA", 1, err, sizeof(err)) != nullptr) return 4;     // cuAprilTags mode: tag36h11 only
  if (node_shell_create(64, 0.22, 4, "tag36h11", "CUDA,,BOGUS", 1, err, sizeof(err)) != nullptr) return 5;
  NodeShellHarness* h = node_shell_create(64, 0.22, 4, "tag25h9", "CPU, CUDA", 1, err, sizeof(err));
  if (!h) { fprintf(stderr, "create: %s\n", err); return 6; }
  std::vector<uint8_t> img(640 * 480 * 3, 128);
  const double K[9] = {500, 0, 320, 0, 500, 240, 0, 0, 1};
  char fid[64];
  std::vector<unsigned char> out(64 * 256);
  NodeShellDetection* o = reinterpret_cast<NodeShellDetection*>(out.data());
  int rc = node_shell_on_frame(h, img.data(), 0, "mono8", 640, 480, 640, K, "cam", 1, 5, 1, 6, o, 64, fid, sizeof(fid), err, sizeof(err));
  if (rc != -1) return 7;   // stamps differ: nothing happens
  rc = node_shell_on_frame(h, img.data(), 0, "yuv422", 640, 480, 1280, K, "cam", 1, 5, 1, 5, o, 64, fid, sizeof(fid), err, sizeof(err));
  if (rc != -2) return 8;   // unsupported encoding throws
  rc = node_shell_on_frame(h, img.data(), 0, "rgb8", 640, 480, 100, K, "cam", 1, 5, 1, 5, o, 64, fid, sizeof(fid), err, sizeof(err));
  if (rc != -2) return 9;   // step smaller than a row
  rc = node_shell_on_frame(h, img.data(), 0, "mono8", 640, 480, 640, K, "cam", 2, 0, 2, 0, o, 64, fid, sizeof(fid), err, sizeof(err));
  printf("valid frame on this machine: %d (%s)\n", rc, rc == -2 ? err : "ok");   // -2 without a HIP device, >= 0 with one
  if (rc == -1) return 10;
  node_shell_destroy(h);
  printf("ok\n");
  return 0;
}
