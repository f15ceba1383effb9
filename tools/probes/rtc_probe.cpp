// Probe: can hiprtc compile the work-group kernel templates for gfx950 (no GPU needed)?
#include <hip/hiprtc.h>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
static std::string slurp(const std::string& p) { std::ifstream f(p); std::stringstream s; s << f.rdbuf(); return s.str(); }
int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "portfft_amd/csrc/";
  const char* names[] = {"stockham_wg.hpp", "butterflies.hpp", "radix_constants.inc"};
  std::vector<std::string> bodies;
  for (auto n : names) bodies.push_back(slurp(dir + n));
  const char* hdrs[3] = {bodies[0].c_str(), bodies[1].c_str(), bodies[2].c_str()};
  const std::string cfg = argc > 2 ? argv[2] : "pfa::wg_cfg<float, pfa::radix_list<15, 10, 8>, 240, 2, 4, 1, 0, 4, 2, 0>";
  std::string src = "#include \"stockham_wg.hpp\"\n";
  const std::string e0 = "pfa::stockham_wg_kernel<" + cfg + ", false>";
  const std::string e1 = "pfa::stockham_wg_kernel<" + cfg + ", true>";
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "pfft_jit.hip", 3, hdrs, names) != HIPRTC_SUCCESS) { puts("create failed"); return 1; }
  hiprtcAddNameExpression(prog, e0.c_str());
  hiprtcAddNameExpression(prog, e1.c_str());
  const char* opts[] = {"--offload-arch=gfx950", "-std=c++17", "-O3", "-ffast-math"};
  auto t0 = std::chrono::steady_clock::now();
  hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
  auto t1 = std::chrono::steady_clock::now();
  size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls);
  std::string log(ls, 0); if (ls) hiprtcGetProgramLog(prog, &log[0]);
  printf("compile: %s in %.2f s\n%s\n", hiprtcGetErrorString(r), std::chrono::duration<double>(t1 - t0).count(), log.substr(0, 3000).c_str());
  if (r != HIPRTC_SUCCESS) return 1;
  const char* low = nullptr; hiprtcGetLoweredName(prog, e0.c_str(), &low); printf("lowered: %s\n", low);
  size_t cs = 0; hiprtcGetCodeSize(prog, &cs); printf("code size %zu\n", cs);
  return 0;
}
