// Host check of csrc/dq_math.h (the arithmetic of the DQB kernels, compiled here for the CPU):
//   dq_math_host rows N K norm_nodes out_mode < binary floats  -> the ROWS form over N rows: prints outputs and gradients
// stdin: q (N*K*4) t (N*K*3) w (N*K) g_rot (N*{9,4,16}) g_t (N*3) as raw float32; stdout: out_rot, out_t, gq, gt, gw raw float32.
// tools/scratch/dq_math_check.py drives it against oracle/dq_ref.py and the reference's goldens.
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../riggs_amd/csrc/dq_math.h"
using namespace riggs;
static std::vector<float> rd(size_t n) { std::vector<float> v(n); if (fread(v.data(), 4, n, stdin) != n) exit(3); return v; }
int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const int N = atoi(argv[2]), K = atoi(argv[3]), nn = atoi(argv[4]), om = atoi(argv[5]);
  const int rw = om == 0 ? 9 : (om == 1 ? 4 : 16);
  std::vector<float> q = rd((size_t)N * K * 4), t = rd((size_t)N * K * 3), w = rd((size_t)N * K), g_rot = rd((size_t)N * rw), g_t = rd((size_t)N * 3);
  std::vector<float> out_rot((size_t)N * rw), out_t((size_t)N * 3), gq(q.size()), gt(t.size()), gw(w.size());
  DqArgs a{};
  a.N = N; a.K = K; a.norm_nodes = nn; a.out_mode = om;
  a.q = q.data(); a.t = t.data(); a.w = w.data(); a.out_rot = out_rot.data(); a.out_t = out_t.data();
  a.g_rot = g_rot.data(); a.g_t = g_t.data(); a.gq = gq.data(); a.gt = gt.data(); a.gw = gw.data();
  for (int n = 0; n < N; n++) {
    if (K <= 2) { dq_row<2, false>(a, n); dq_row<2, true>(a, n); }
    else if (K <= 4) { dq_row<4, false>(a, n); dq_row<4, true>(a, n); }
    else { dq_row<8, false>(a, n); dq_row<8, true>(a, n); }
  }
  fwrite(out_rot.data(), 4, out_rot.size(), stdout); fwrite(out_t.data(), 4, out_t.size(), stdout);
  fwrite(gq.data(), 4, gq.size(), stdout); fwrite(gt.data(), 4, gt.size(), stdout); fwrite(gw.data(), 4, gw.size(), stdout);
  return 0;
}
