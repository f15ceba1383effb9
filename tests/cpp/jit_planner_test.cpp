// Runtime-specialisation planner (portfft_amd/csrc/jit_planner.cpp) on the host: invariants of the chosen kernel
// parameters for every length, and hiprtc compilation of a few of them for gfx950 (no device needed).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../portfft_amd/csrc/jit.hpp"

static int fails = 0;
#define EXPECT(c, ...)                 \
  do {                                 \
    if (!(c)) {                        \
      ++fails;                         \
      std::printf("FAIL %s: ", #c);    \
      std::printf(__VA_ARGS__);        \
      std::printf("\n");               \
    }                                  \
  } while (0)

/// prime factors up to the wavefront size (jit_planner.cpp: jit_max_prime)
static bool smooth31(long long n) {
  for (int p = 2; p <= 61; ++p) {
    while (n % p == 0) n /= p;
  }
  return n == 1;
}

int main(int argc, char** argv) {
  const size_t max_lds = 160 * 1024;
  long long planned[2] = {0, 0};
  for (int prec = 0; prec < 2; ++prec) {
    const int es = prec ? 16 : 8;
    for (long long n = 2; n <= 21000; ++n) {
      pfa::wg_params p;
      const bool ok = pfa::choose_spec_params(prec, n, max_lds, &p);
      if (!smooth31(n) || static_cast<size_t>(n) * es > max_lds) {
        EXPECT(!ok, "n=%lld prec=%d should not be planned", n, prec);
        continue;
      }
      EXPECT(ok, "n=%lld prec=%d not planned", n, prec);
      if (!ok) continue;
      ++planned[prec];
      long long prod = 1;
      for (int r : p.radices) {
        prod *= r;
        EXPECT(r >= 2 && r <= 61 && (r <= 32 || (r % 2 != 0 && r % 3 != 0 && r % 5 != 0 && r % 7 != 0)),
               "n=%lld radix %d", n, r);
      }
      EXPECT(prod == n, "n=%lld product %lld", n, prod);
      EXPECT(p.radices.size() <= 6, "n=%lld passes", n);
      EXPECT(p.wg >= 1 && p.wg <= 1024 && p.fpw >= 1 && p.wg % p.fpw == 0, "n=%lld wg=%d fpw=%d", n, p.wg, p.fpw);
      EXPECT(pfa::spec_lds_bytes(p) <= max_lds, "n=%lld lds=%zu", n, pfa::spec_lds_bytes(p));
      EXPECT(p.regs <= 64, "n=%lld regs=%d", n, p.regs);
      const int tpf = p.wg / p.fpw;
      for (int r : p.radices) {
        const long long bpt = (n / r + tpf - 1) / tpf;
        EXPECT(bpt * r <= p.regs, "n=%lld radix %d bpt %lld regs %d", n, r, bpt, p.regs);
      }
      pfa::wg_params q;
      if (pfa::choose_strided_params(prec, n, 1000, max_lds, &q)) {
        long long pq = 1;
        for (int r : q.radices) pq *= r;
        EXPECT(pq == n && (q.radices.size() >= 2 || n <= 61), "strided n=%lld", n);  // one lane per FFT up to the largest radix
        EXPECT(q.wg >= 64 && q.wg <= 1024 && q.wg % q.fpw == 0, "strided n=%lld wg=%d fpw=%d", n, q.wg, q.fpw);
        // (one FFT per work-group -- lengths beyond the generic tier's two images -- may use all of the LDS)
        EXPECT(static_cast<size_t>(n) * q.fpw * es <= (q.fpw == 1 ? max_lds : 128 * 1024), "strided n=%lld lds", n);
        EXPECT(q.fpw > 1 || static_cast<size_t>(n) * 2 * es > max_lds, "strided n=%lld: one FFT per work-group", n);
      }
      // a requested group width (four-step half pairs, plan_global.cpp) is honoured exactly or refused
      for (int want : {8, 16}) {
        pfa::wg_params w;
        if (n > 32 && pfa::choose_strided_params(prec, n, 1024, max_lds, &w, false, want)) {
          EXPECT(w.fpw == want && w.wg % want == 0 && w.wg >= 64 && w.wg <= 1024, "strided n=%lld want %d: fpw=%d wg=%d",
                 n, want, w.fpw, w.wg);
          EXPECT(static_cast<size_t>(n) * want * es <= 128 * 1024, "strided n=%lld want %d lds", n, want);
        }
      }
    }
  }
  std::printf("planned %lld fp32 and %lld fp64 lengths\n", planned[0], planned[1]);
  // choices the measurements in profiles/r1_notes.md rest on
  pfa::wg_params p;
  pfa::choose_spec_params(0, 243, max_lds, &p);
  EXPECT(p.radices.size() == 3 && p.radices[0] == 9, "243 -> 9,9,3");
  pfa::choose_spec_params(0, 4096, max_lds, &p);
  EXPECT(p.radices.size() == 3 && p.radices[0] == 16 && p.wg == 256 && p.fpw == 1, "4096 -> 16,16,16 / 256 lanes");
  // two-pass 2-D plan, pass 1: the planner's invariants for every row length it accepts
  for (int prec = 0; prec < 2; ++prec) {
    int accepted = 0;
    for (long long n1 = 32; n1 <= 8192; ++n1) {
      for (long long n0 : {64ll, 1000ll, 1026ll}) {
        pfa::wg_params q;
        if (!pfa::choose_rows2d_params(prec, n1, n0, max_lds, &q)) continue;
        ++accepted;
        long long prod = 1;
        for (int r : q.radices) prod *= r;
        const long long nbl = n1 / q.radices.back();
        EXPECT(prod == n1 && q.radices.size() >= 2, "rows2d %lld: radices multiply to n1", n1);
        EXPECT(n0 % q.fpw == 0 && n0 / q.fpw >= 2, "rows2d %lld x %lld: column radix %d divides n0", n1, n0, q.fpw);
        EXPECT(nbl % q.wg == 0 && q.wg % q.fpw == 0 && q.wg <= 1024, "rows2d %lld: lanes %d", n1, q.wg);
        EXPECT(q.regs <= (prec ? 16 : 32), "rows2d %lld: %d elements per lane", n1, q.regs);
        EXPECT(pfa::spec_lds_bytes(q) <= max_lds, "rows2d %lld: LDS", n1);
      }
    }
    EXPECT(accepted > 500, "rows2d planner accepts %d (n1, n0) pairs", accepted);
    // the caller's mask of usable column radices is honoured (3000 x 1000: radix 2 would leave 1500-point columns)
    pfa::wg_params q;
    EXPECT(pfa::choose_rows2d_params(prec, 1000, 3000, max_lds, &q, 4) && q.fpw == 4, "rows2d 3000 x 1000, mask 4");
    EXPECT(!pfa::choose_rows2d_params(prec, 1000, 3000, max_lds, &q, 8), "rows2d 3000 x 1000: no radix-8 plan");
  }
  // register-resident packed kernel (stockham_wg_hx.hpp): lengths beyond the LDS whose transform fits the registers of one
  // work-group -- invariants of every plan the planner hands out
  {
    long long planned_hx[2] = {0, 0};
    pfa::wg_params p0;
    for (int prec = 0; prec < 2; ++prec) {
      const int es = prec ? 16 : 8;
      for (long long n = (prec ? 10241 : 19000); n <= (prec ? 21000 : 42000); ++n) {
        pfa::wg_params p;
        if (!pfa::choose_hx_params(prec, n, max_lds, &p)) continue;
        ++planned_hx[prec];
        long long prod = 1;
        int regs = 0;
        for (int r : p.radices) {
          prod *= r;
          EXPECT(r >= 2 && r <= 32, "hx n=%lld radix %d", n, r);
          regs = std::max<int>(regs, static_cast<int>((n / r + p.wg - 1) / p.wg) * r);
        }
        EXPECT(prod == n && p.radices.size() >= 2 && p.radices.size() <= 4, "hx n=%lld radices", n);
        if (p.hx_pair != 0) continue;  // (checked below)
        EXPECT(p.fpw == 1 && p.staged == 0 && p.wg % 64 == 0 && p.wg >= 512 && p.wg <= 1024, "hx n=%lld lanes %d", n, p.wg);
        EXPECT(pfa::hx_lds_bytes(p) <= max_lds && static_cast<size_t>(n) * es > 152 * 1024, "hx n=%lld LDS", n);
        const int budget = 512 / ((p.wg / 64 + 3) / 4);
        EXPECT(regs * (prec ? 4 : 2) < budget && p.regs == regs, "hx n=%lld: %d elements per lane of %d lanes", n, regs, p.wg);
      }
    }
    std::printf("hx planner: %lld fp32 and %lld fp64 lengths\n", planned_hx[0], planned_hx[1]);
    EXPECT(planned_hx[0] > 300 && planned_hx[1] > 150, "hx planner coverage");
    EXPECT(!pfa::choose_hx_params(0, 40960, max_lds, &p0) , "40960: 80 values per lane do not pay");
    pfa::wg_params p;
    // 80 ... 152 KiB: TWO register-resident work-groups per CU (half the LDS, half the registers each), three passes
    // behind a first radix >= 15 with a conflict-free scatter (odd, or even and padded with its own period) -- or the
    // packed planner's LDS-resident kernel
    long long pairs[2] = {0, 0};
    for (int prec = 0; prec < 2; ++prec) {
      const int es = prec ? 16 : 8;
      for (long long n = 80 * 1024 / es - 100; n <= (prec ? 10240 : 19000); ++n) {
        pfa::wg_params q;
        if (!pfa::choose_hx_params(prec, n, max_lds, &q)) continue;
        ++pairs[prec];
        EXPECT(static_cast<size_t>(n) * es > 80 * 1024 && q.hx_pair == 1, "hx pair n=%lld", n);
        EXPECT(q.radices.size() <= 3 && q.radices[0] >= 15 && q.pads == (q.radices[0] % 2 == 0 ? q.radices[0] : 0),
               "hx pair n=%lld: three passes, an even first radix padded", n);
        EXPECT(q.wg >= 256 && q.wg <= 512 && q.wg % 64 == 0 && 2 * pfa::hx_lds_bytes(q) <= max_lds, "hx pair n=%lld: lanes %d", n, q.wg);
        const int budget = 512 / ((2 * (q.wg / 64) + 3) / 4);
        EXPECT(q.regs * (prec ? 4 : 2) < budget && q.occ == (2 * (q.wg / 64) + 3) / 4, "hx pair n=%lld: %d values per lane", n, q.regs);
      }
    }
    std::printf("hx planner: %lld fp32 and %lld fp64 lengths as pairs\n", pairs[0], pairs[1]);
    EXPECT(pairs[0] > 100 && pairs[1] > 50, "hx pair coverage");
    EXPECT(pfa::choose_hx_params(0, 16384, max_lds, &p) && p.hx_pair == 1 && p.wg == 512, "fp32 16384: two work-groups per CU");
    EXPECT(!pfa::choose_hx_params(0, 18000, max_lds, &p), "fp32 18000: no three-pass pair plan, LDS-resident");
    EXPECT(!pfa::choose_hx_params(0, 8192, max_lds, &p), "fp32 8192: the LDS-resident kernel has two work-groups per CU itself");
    EXPECT(pfa::choose_hx_params(0, 20480, max_lds, &p), "fp32 20480: the top of the LDS range goes register-resident");
    EXPECT(!pfa::choose_hx_params(1, 10240, max_lds, &p), "fp64 10240 stays LDS-resident");
    EXPECT(pfa::choose_hx_params(1, 6144, max_lds, &p) && p.hx_pair == 1 && p.wg == 256, "fp64 6144: two work-groups per CU");
    EXPECT(!pfa::choose_hx_params(0, 65536, max_lds, &p), "65536 does not fit the registers");
    EXPECT(pfa::choose_hx_params(0, 32768, max_lds, &p) && p.radices.size() == 3, "fp32 32768 in three passes");
    EXPECT(pfa::choose_hx_params(1, 16384, max_lds, &p) && p.radices.size() <= 4, "fp64 16384");
  }
  {
    // register-resident strided plans (strided_hx_candidates): only for groups that would sit alone on their CU, a half
    // image that lets the planned number of work-groups share the CU's LDS, lanes inside the register budget
    long long planned[2] = {0, 0};
    for (int prec = 0; prec < 2; ++prec) {
      const int es = prec ? 16 : 8;
      for (long long n : {104LL, 256LL, 512LL, 600LL, 625LL, 660LL, 720LL, 768LL, 800LL, 1000LL, 1024LL, 1100LL, 1536LL, 1944LL, 2000LL, 2048LL}) {
        pfa::wg_params b;
        if (!pfa::choose_strided_params(prec, n, 4096, max_lds, &b, false, prec ? 8 : 16)) continue;
        const std::vector<pfa::wg_params> c = pfa::strided_hx_candidates(b, max_lds);
        const size_t full = static_cast<size_t>(n) * b.fpw * es;
        EXPECT(full > 80 * 1024 || c.empty(), "strided hx n=%lld: a group of %zu bytes has two work-groups per CU already", n, full);
        for (const pfa::wg_params& q : c) {
          ++planned[prec];
          long long prod = 1;
          for (int r : q.radices) prod *= r;
          EXPECT(prod == n && q.fpw == b.fpw && q.hx_strided >= 2 && q.hx_strided <= 4, "strided hx n=%lld: same group, 2 ... 4 per CU", n);
          EXPECT(static_cast<size_t>(q.hx_strided) * pfa::strided_hx_lds_bytes(q) <= max_lds, "strided hx n=%lld: LDS of %d work-groups", n, q.hx_strided);
          EXPECT(pfa::strided_hx_lds_bytes(q) < full * 3 / 4, "strided hx n=%lld: the image is about half of the group", n);
          const int waves = (q.wg + 63) / 64, wps = (q.hx_strided * waves + 3) / 4;
          EXPECT(q.wg <= 1024 && q.wg % q.fpw == 0 && q.occ == wps && q.regs * (prec ? 4 : 2) < 512 / wps,
                 "strided hx n=%lld: %d lanes, %d values per lane", n, q.wg, q.regs);
          EXPECT(prec == 1 || q.hx_strided * waves >= 14, "strided hx n=%lld: fp32 plans keep 14 waves on the CU", n);
        }
      }
    }
    std::printf("strided hx planner: %lld fp32 and %lld fp64 candidate plans\n", planned[0], planned[1]);
    EXPECT(planned[0] >= 4 && planned[1] >= 3, "strided hx coverage");
    {
      pfa::wg_params b;
      EXPECT(pfa::choose_strided_params(0, 1000, 4096, max_lds, &b, false, 16) && pfa::strided_hx_candidates(b, max_lds).empty(),
             "fp32 1000 x 16: 16000 values do not fit two work-groups' registers beside ~62 of overhead -- LDS-resident");
    }
  }
  {
    // WIDE groups (choose_strided_wide_base + strided_hx_candidates(..., wide)): a full-width group of a length beyond the LDS, one
    // work-group per CU on a half image that fits the CU's LDS, values + overhead inside the wave's register budget
    long long planned[2] = {0, 0};
    for (int prec = 0; prec < 2; ++prec) {
      const int es = prec ? 16 : 8, fpw = prec ? 8 : 16;
      for (long long n : {1024LL, 1280LL, 1536LL, 1944LL, 2000LL, 2048LL, 3072LL, 4096LL}) {
        pfa::wg_params b;
        if (!pfa::choose_strided_wide_base(prec, n, fpw, &b)) continue;
        const std::vector<pfa::wg_params> c = pfa::strided_hx_candidates(b, max_lds, true);
        const size_t full = static_cast<size_t>(n) * fpw * es;
        EXPECT(full > 128 * 1024 || c.empty(), "wide strided hx n=%lld: a group of %zu bytes fits the LDS", n, full);
        EXPECT(n <= 2048 || c.empty(), "wide strided hx n=%lld: more than 32 values per lane on 1024 lanes", n);
        for (const pfa::wg_params& q : c) {
          ++planned[prec];
          long long prod = 1;
          for (int r : q.radices) prod *= r;
          EXPECT(prod == n && q.fpw == fpw && q.hx_strided == 1, "wide strided hx n=%lld: the same group, one per CU", n);
          EXPECT(pfa::strided_hx_lds_bytes(q) + 4096 <= max_lds, "wide strided hx n=%lld: LDS", n);
          const int waves = (q.wg + 63) / 64, wps = (waves + 3) / 4;
          EXPECT(q.wg <= 1024 && q.wg % q.fpw == 0 && q.occ == wps && q.regs * (prec ? 4 : 2) < 512 / wps,
                 "wide strided hx n=%lld: %d lanes, %d values per lane", n, q.wg, q.regs);
        }
      }
    }
    {
      // the planner's own radices of 2000 (10.8.5.5) hold 40 values per lane on 64 lanes per transform: the base takes the set with the
      // narrowest widest pass instead (16.5.5.5: 35)
      pfa::wg_params b;
      EXPECT(pfa::choose_strided_wide_base(0, 2000, 16, &b) && b.radices == std::vector<int>({16, 5, 5, 5}), "wide base of 2000: 16.5.5.5");
      const std::vector<pfa::wg_params> c = pfa::strided_hx_candidates(b, max_lds, true);
      EXPECT(!c.empty() && c[0].regs == 35 && c[0].wg >= 960, "wide plan of 2000: 35 values per lane on ~1000 lanes");
      // a group that fits the LDS is no wide group; neither is one of more than 36 values per lane
      EXPECT(pfa::choose_strided_wide_base(0, 1000, 16, &b) && pfa::strided_hx_candidates(b, max_lds, true).empty(), "1000 x 16 fits the LDS");
      EXPECT(pfa::choose_strided_wide_base(0, 4096, 16, &b) && pfa::strided_hx_candidates(b, max_lds, true).empty(), "4096 x 16: 64 values per lane");
    }
    std::printf("wide strided hx planner: %lld fp32 and %lld fp64 candidate plans\n", planned[0], planned[1]);
    EXPECT(planned[0] >= 3 && planned[1] >= 4, "wide strided hx coverage");
  }
  if (argc > 1 && std::string(argv[1]) == "compile") {
    for (auto c : std::vector<std::pair<int, long long>>{{0, 2048}, {1, 2048}, {0, 1536}}) {
      pfa::wg_params b;
      EXPECT(pfa::choose_strided_wide_base(c.first, c.second, c.first ? 8 : 16, &b), "wide base %lld", c.second);
      const std::vector<pfa::wg_params> cand = pfa::strided_hx_candidates(b, max_lds, true);
      EXPECT(!cand.empty(), "wide strided hx plan %lld", c.second);
      if (cand.empty()) continue;
      size_t bytes = 0;
      std::string why;
      const bool built = pfa::jit_compile_only(cand[0], 11, "gfx950", &bytes, &why);
      EXPECT(built && bytes > 1000, "hiprtc wide strided hx n=%lld: %s", c.second, why.c_str());
      std::printf("hiprtc wide n=%lld %s: %zu bytes\n", c.second, pfa::wg_cfg_type_name(cand[0]).c_str(), bytes);
    }
    for (auto c : std::vector<std::pair<int, long long>>{{0, 660}, {1, 660}, {0, 768}}) {
      pfa::wg_params b;
      EXPECT(pfa::choose_strided_params(c.first, c.second, 4096, max_lds, &b, false, c.first ? 8 : 16), "strided plan %lld", c.second);
      const std::vector<pfa::wg_params> cand = pfa::strided_hx_candidates(b, max_lds);
      EXPECT(!cand.empty(), "strided hx plan %lld", c.second);
      for (int kind : {11, 12}) {
        if (cand.empty()) break;
        size_t bytes = 0;
        std::string why;
        const bool built = pfa::jit_compile_only(cand[0], kind, "gfx950", &bytes, &why);
        EXPECT(built && bytes > 1000, "hiprtc strided hx n=%lld kind=%d: %s", c.second, kind, why.c_str());
        std::printf("hiprtc n=%lld kind=%d %s x%d per CU: %zu bytes\n", c.second, kind, pfa::wg_cfg_type_name(cand[0]).c_str(),
                    cand[0].hx_strided, bytes);
      }
    }
    for (auto c : std::vector<std::pair<int, long long>>{{0, 24576}, {0, 30000}, {1, 12000}, {1, 15000}}) {
      pfa::wg_params q;
      EXPECT(pfa::choose_hx_params(c.first, c.second, max_lds, &q), "hx plan %lld", c.second);
      for (int kind : {8, 9}) {
        size_t bytes = 0;
        std::string why;
        const bool built = pfa::jit_compile_only(q, kind, "gfx950", &bytes, &why);
        EXPECT(built && bytes > 1000, "hiprtc hx n=%lld kind=%d: %s", c.second, kind, why.c_str());
        std::printf("hiprtc n=%lld kind=%d %s: %zu bytes\n", c.second, kind, pfa::wg_cfg_type_name(q).c_str(), bytes);
      }
    }
    struct { int prec; long long n; int kind; } cases[] = {{0, 1200, 0}, {1, 625, 1}, {0, 30, 0}, {0, 120, 2}, {1, 250, 3},
                                                            {0, 1000, 4}, {1, 768, 4},
                                                            // forms of the three-stage / tiled plans that exist only at run time
                                                            {0, 128, 5}, {1, 1024, 6}, {0, 1024, 7}};
    for (auto& c : cases) {
      pfa::wg_params q;
      const bool ok = c.kind < 2 ? pfa::choose_spec_params(c.prec, c.n, max_lds, &q)
                      : c.kind == 4 ? pfa::choose_rows2d_params(c.prec, c.n, 1200, max_lds, &q)
                                    : pfa::choose_strided_params(c.prec, c.n, 1024, max_lds, &q, false,
                                                                 c.kind >= 6 ? (c.prec ? 8 : 16) : 0);
      EXPECT(ok, "plan %lld", c.n);
      size_t bytes = 0;
      std::string why;
      const bool built = pfa::jit_compile_only(q, c.kind, "gfx950", &bytes, &why);
      EXPECT(built && bytes > 1000, "hiprtc n=%lld kind=%d: %s", c.n, c.kind, why.c_str());
      std::printf("hiprtc n=%lld kind=%d %s: %zu bytes\n", c.n, c.kind, pfa::wg_cfg_type_name(q).c_str(), bytes);
    }
  }
  // fused N-D tier: fits-in-LDS rule, pass products, and (with "compile") the nd kernel templates under hiprtc
  {
    struct { int prec; std::vector<long long> dims; bool ok; } shapes[] = {
        {0, {64, 64}, true}, {1, {16, 16, 16}, true}, {0, {30, 50}, true}, {0, {128, 128}, true},
        {1, {128, 128}, false}, {0, {67, 8}, false},  {0, {4, 1, 8}, true}, {0, {256, 128}, false}};
    for (auto& sh : shapes) {
      pfa::nd_kernel nk;
      const bool ok = pfa::choose_nd_params(sh.prec, sh.dims, max_lds, &nk);
      EXPECT(ok == sh.ok, "nd shape %lldx%lld.. planned=%d", sh.dims[0], sh.dims[1], int(ok));
      if (!ok) continue;
      long long tot = 1;
      for (size_t d = 0; d < sh.dims.size(); ++d) {
        long long pd = 1;
        for (int r : nk.radices[d]) pd *= r;
        EXPECT(pd == sh.dims[d], "nd dim product");
        tot *= sh.dims[d];
      }
      EXPECT(nk.k.n == tot && nk.k.wg % nk.k.fpw == 0 && nk.k.wg <= 1024 && nk.k.lds_bytes <= 160 * 1024,
             "nd kernel shape n=%d wg=%d fpw=%d lds=%zu", nk.k.n, nk.k.wg, nk.k.fpw, nk.k.lds_bytes);
      if (argc > 1 && std::string(argv[1]) == "compile" && tot >= 512 && tot <= 4096) {
        size_t bytes = 0;
        std::string why;
        const bool built = pfa::jit_compile_only_nd(nk, sh.prec == 1, "gfx950", &bytes, &why);
        EXPECT(built && bytes > 1000, "hiprtc nd: %s", why.c_str());
        std::printf("hiprtc nd %s: %zu bytes\n", pfa::nd_cfg_type_name(nk).c_str(), bytes);
      }
    }
  }
  // measured planning (PFFT_PLAN_MEASURE): the candidates start with the planner's own choice, each is a valid plan of
  // the length, and a recorded choice (process table / cache directory) is found again and honoured by the planner
  {
    for (long long n : {3000ll, 6000ll, 10080ll, 625ll, 15625ll}) {
      pfa::wg_params base;
      EXPECT(pfa::choose_spec_params(0, n, max_lds, &base), "n=%lld planned", n);
      const auto cands = pfa::spec_radix_candidates(0, n, max_lds);
      EXPECT(!cands.empty() && cands[0] == base.radices, "n=%lld first candidate is the static choice", n);
      EXPECT(cands.size() >= 2 && cands.size() <= 14, "n=%lld %zu candidates", n, cands.size());
      for (const auto& r : cands) {
        long long prod = 1;
        for (int x : r) prod *= x;
        pfa::wg_params q;
        EXPECT(prod == n && pfa::choose_spec_params(0, n, max_lds, &q, &r) && q.radices == r &&
                   pfa::spec_lds_bytes(q) <= max_lds,
               "n=%lld candidate", n);
      }
    }
    const std::vector<int> pick = {24, 25, 10};
    EXPECT(pfa::plan_choice_lookup("gfx000", 0, 6000).empty(), "no record yet");
    pfa::plan_choice_store("gfx000", 0, 6000, pick);
    EXPECT(pfa::plan_choice_lookup("gfx000", 0, 6000) == pick, "recorded choice found");
    EXPECT(pfa::plan_choice_lookup("gfx000", 1, 6000).empty() && pfa::plan_choice_lookup("gfx000", 0, 3000).empty(),
           "records are per precision and length");
    pfa::wg_params q;
    EXPECT(pfa::choose_spec_params(0, 6000, max_lds, &q, &pick) && q.radices == pick, "forced radices honoured");
    const std::vector<int> bad = {24, 25, 11};
    EXPECT(pfa::choose_spec_params(0, 6000, max_lds, &q, &bad) && q.radices != bad, "a sequence of another length is ignored");
    {  // lanes per transform behind the radices: "0, lanes"
      const std::vector<int> with_lanes = {16, 61, 0, 16};
      pfa::wg_params w;
      EXPECT(pfa::choose_spec_params(0, 976, max_lds, &w, &with_lanes) && w.radices == std::vector<int>({16, 61}) &&
                 w.wg == 16 * w.fpw,
             "forced lanes honoured: wg %d fpw %d", w.wg, w.fpw);
      pfa::plan_choice_store("gfx000", 0, 976, with_lanes);
      EXPECT(pfa::plan_choice_lookup("gfx000", 0, 976) == with_lanes, "a record with lanes is read back");
      pfa::plan_choice_store("gfx000", 0, 976, {});
    }
    pfa::plan_choice_store("gfx000", 0, 6000, {});  // forget it again: the cache directory may be a shared one
    EXPECT(pfa::plan_choice_lookup("gfx000", 0, 6000).empty(), "record forgotten");
  }
  {  // the tuned table shipped with the library (tuned_gfx950.inc): every entry is a sequence the planner accepts
    int entries = 0;
    for (int prec = 0; prec < 2; ++prec) {
      for (long long n = 2; n <= 1000000; n += (n < 21000 ? 1 : 4)) {
        const std::vector<int> c = pfa::builtin_choice("gfx950", prec, n, false);
        const std::vector<int> sp = pfa::builtin_choice("gfx950", prec, n, true);
        if (!c.empty()) {
          ++entries;
          std::vector<int> rad = c;  // (an entry may end in "0, lanes")
          int lanes = 0;
          if (rad.size() >= 3 && rad[rad.size() - 2] == 0) {
            lanes = rad.back();
            rad.resize(rad.size() - 2);
          }
          long long prod = 1;
          for (int x : rad) prod *= x;
          pfa::wg_params q;
          EXPECT(prod == n && pfa::choose_spec_params(prec, n, max_lds, &q, &c) && q.radices == rad &&
                     (lanes == 0 || q.wg == lanes * q.fpw),
                 "tuned radices of n=%lld prec=%d", n, prec);
        }
        if (!sp.empty()) {
          ++entries;
          EXPECT(sp.size() == 2 && static_cast<long long>(sp[0]) * sp[1] == n, "tuned split of n=%lld", n);
        }
        EXPECT(pfa::builtin_choice("gfx942", prec, n, false).empty() && pfa::builtin_choice("gfx942", prec, n, true).empty(),
               "the table is for gfx950 only");
      }
    }
    setenv("PFFT_NO_TUNED_TABLE", "1", 1);
    EXPECT(pfa::builtin_choice("gfx950", 0, 6000, false).empty() && pfa::builtin_choice("gfx950", 0, 100000, true).empty(),
           "PFFT_NO_TUNED_TABLE=1 turns the table off");
    unsetenv("PFFT_NO_TUNED_TABLE");
    std::printf("tuned table: %d entries checked\n", entries);
  }
  {
    long long compiled = 0, from_disk = 0;
    pfa::jit_stats(&compiled, &from_disk);
    std::printf("jit stats: compiled %lld from_disk %lld\n", compiled, from_disk);
  }
  if (fails == 0) std::printf("jit planner OK\n");
  return fails == 0 ? 0 : 1;
}
