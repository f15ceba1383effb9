// plan_t: the stage list of one direction -- 1-D, fused N-D, the two-pass 2-D plan, per-dimension passes (plan.hpp).
#include "plan.hpp"

namespace pfa {

const rows2d_kernel* plan_t::find_rows2d_registered(long long n1, long long n0, int policy, bool split) const {
  if (kn.two_pass_2d_off) return nullptr;  // experiments / parity A-B: rows, then full-length columns
  int count = 0;
  const rows2d_kernel* k = rows2d_kernels(&count);
  for (int i = 0; i < count; ++i) {
    if (k[i].precision == desc.precision && k[i].n == n1 && k[i].lds_bytes <= max_lds && n0 % k[i].rc == 0 &&
        n0 / k[i].rc >= 2 && k[i].policy == policy && (!split || k[i].launch_split != nullptr)) {
      return &k[i];
    }
  }
  return nullptr;
}

const rows2d_kernel* plan_t::find_rows2d(long long n1, long long n0, int policy, bool split) {
  if (kn.two_pass_2d_off) return nullptr;
  if (const rows2d_kernel* k = find_rows2d_registered(n1, n0, policy, split)) return k;
  // other row lengths / column counts: the same template instantiated at commit (jit.cpp) -- when the full-length
  // column pass it replaces would move segments below 256 bytes (measured, tools/perf_2d.py: 1080 x 1920 +27 %,
  // 1536^2 +29 %, 3000 x 1000 2.6x, 4096^2 2.1x; with 256-byte column segments available, 384^2 ... 960^2, the
  // runtime-planned first pass does not beat rows + columns)
  const int col_fpw = strided_fpw(n0, n1);
  if (col_fpw > 0 && static_cast<size_t>(col_fpw) * elem_bytes() >= 256) return nullptr;
  // column radices whose remaining n0 / rc points one full-width column pass can take (plan_1d would otherwise
  // answer with two column stages through scratch, and the two-pass plan would be dropped: 3000 x 1000 with rc 2);
  // failing that, any radix with a column kernel at all
  const int full_fpw = desc.precision == PFFT_PRECISION_F64 ? 8 : 16;
  int wide_mask = 0, any_mask = 0;
  for (int rc : {8, 4, 2}) {
    if (n0 % rc != 0 || n0 / rc < 2) continue;
    const int fpw = strided_fpw(n0 / rc, rc * n1);
    if (fpw >= full_fpw) wide_mask |= rc;
    if (fpw > 0) any_mask |= rc;
  }
  const int rc_mask = wide_mask != 0 ? wide_mask : any_mask;
  if (rc_mask == 0) return nullptr;
  std::string why;
  const rows2d_kernel* jk = jit_rows2d_kernel(desc.precision, n1, n0, max_lds, &why, policy, split ? 1 : 0, rc_mask);
  if (jk == nullptr) jit_note("rows2d", n1, why);
  return jk;
}

stage plan_t::make_rows2d_stage(const rows2d_kernel* k, long long nmat, long long n0, long long in_off,
                                long long out_off, int backward) {
  stage s;
  s.rows2d = k;
  s.n = k->n;
  s.in_buf = BUF_IN;
  s.out_buf = BUF_OUT;
  s.in_offset = in_off;
  s.out_offset = out_off;
  s.count = nmat * n0;
  s.backward = backward;
  s.lds_bytes = k->lds_bytes;
  s.alias_scratch = 1;
  s.ra.tw = upload_twiddles(std::vector<int>(k->radices, k->radices + k->n_radices));
  s.ra.twc = upload_unit_roots(n0);
  s.ra.nmat = nmat;
  s.ra.n0 = static_cast<int>(n0);
  const bool split = desc.complex_storage == PFFT_SPLIT_COMPLEX;
  for (int d = 0; d < 2 && k->launch != nullptr; ++d) {
    if (k->lds_bytes > 48 * 1024) {
      hip_check(hipFuncSetAttribute((split ? k->fn_split : k->fn)[d], hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(k->lds_bytes)),
                "hipFuncSetAttribute");
    }
  }
  s.grid = persistent_grid(k->launch != nullptr ? (split ? k->fn_split : k->fn)[backward] : nullptr, k->mfn[backward],
                           k->wg, k->lds_bytes, nmat * (n0 / k->rc), k->groups_per_wg);
  return s;
}

void plan_t::build_direction(int direction) {
  std::vector<stage>& st = stages[direction];
  const int inv = direction == PFFT_FORWARD ? PFFT_BACKWARD : PFFT_FORWARD;
  const view_t vin = view_of(desc, direction), vout = view_of(desc, inv);
  bool packed = layout_of(desc, direction) == PFFT_LAYOUT_PACKED && layout_of(desc, inv) == PFFT_LAYOUT_PACKED;
  // (ONE 1-D transform with unit element strides is packed data whatever its distances say -- a batch-interleaved descriptor
  //  with a batch of one: tools/fuzz.py seed 97 found fp32 N = 13038 in that shape `unsupported`, its length beyond the strided
  //  tier's and the GLOBAL tier packed-only)
  if (desc.rank == 1 && desc.number_of_transforms == 1 && vin.n_strides == 1 && vout.n_strides == 1 && vin.strides[0] == 1 &&
      vout.strides[0] == 1) {
    packed = true;
  }
  const double scale = direction == PFFT_FORWARD ? desc.forward_scale : desc.backward_scale;
  const int backward = direction == PFFT_BACKWARD ? 1 : 0;
  const long long B = static_cast<long long>(desc.number_of_transforms);
  const int rank = desc.rank;
  const long long total = static_cast<long long>(flattened_length(desc));
  const bool record = direction == PFFT_FORWARD;
  if (rank == 1) {
    const long long n = static_cast<long long>(desc.lengths[0]);
    addressing ia{static_cast<long long>(vin.offset), static_cast<long long>(vin.strides[0]),
                  static_cast<long long>(vin.distance), 0};
    addressing oa{static_cast<long long>(vout.offset), static_cast<long long>(vout.strides[0]),
                  static_cast<long long>(vout.distance), 0};
    plan_1d(st, n, B, B, BUF_IN, ia, BUF_OUT, oa, packed, scale, backward, record ? &info.dims[0] : nullptr);
    return;
  }
  // N-D, packed (validated): contiguous dimension first, then every outer dimension in place on the output,
  // as strided FFTs (reference: dispatch_dimensions, committed_descriptor_impl.hpp:923-948; there one launch per
  // (batch, outer index), here one launch per dimension).
  // ... unless a suffix of the dimensions fits LDS: one fused launch (stockham_nd.hpp) does lengths[s..rank) for
  // every index of the dimensions before it -- the whole transform when s == 0 -- and only lengths[0..s) remain
  // as strided passes
  int fused_from = rank;  // first dimension covered by the fused stage
  for (int s0 = 0; s0 + 2 <= rank && fused_from == rank; ++s0) {
    const std::vector<long long> dims(desc.lengths + s0, desc.lengths + rank);
    std::string why;
    const nd_kernel* nk =
        jit_nd_kernel(desc.precision, dims, desc.complex_storage == PFFT_SPLIT_COMPLEX, max_lds, &why);
    if (nk == nullptr) continue;
    long long outer = 1;
    for (int i = 0; i < s0; ++i) outer *= static_cast<long long>(desc.lengths[i]);
    st.push_back(make_spec_stage(&nk->k, B * outer, BUF_IN, static_cast<long long>(vin.offset), BUF_OUT,
                                 static_cast<long long>(vout.offset), scale, backward, upload_nd_twiddles(*nk)));
    for (int i = s0; record && i < rank; ++i) {
      pfft_dim_info_t& di = info.dims[i];
      const std::vector<int>& r = nk->radices[static_cast<size_t>(i - s0)];
      di.length = desc.lengths[i];
      di.tier = PFFT_TIER_WORKGROUP;
      di.n_factors = static_cast<int>(std::min<size_t>(r.size(), PFFT_MAX_FACTORS));
      for (int f = 0; f < di.n_factors; ++f) di.factors[f] = r[static_cast<size_t>(f)];
      di.workgroup_size = nk->k.wg;
      di.ffts_per_workgroup = nk->k.fpw;
      di.lds_bytes = nk->k.lds_bytes;
    }
    fused_from = s0;
  }
  if (fused_from == 0) return;
  long long inner = 1;
  // Two-pass plan for the last two dimensions (stockham_rows2d.hpp): pass 1 = whole rows + the first radix-RC
  // butterfly of the columns (contiguous rows on both sides), pass 2 = the remaining (n0 / RC)-point column FFTs
  // as a batch-interleaved transform over RC * n1 adjacent columns.  C5 (fp32 1024 x 1024 x 256): 1.57 -> 1.40 ms.
  if (fused_from == rank) {
    const bool split = desc.complex_storage == PFFT_SPLIT_COMPLEX;
    const long long n0 = static_cast<long long>(desc.lengths[rank - 2]);
    const long long n1 = static_cast<long long>(desc.lengths[rank - 1]);
    const long long nmat = B * (total / (n0 * n1));
    // chunk of matrices whose intermediate fits the Infinity Cache (cache_chunk_bytes): pass 1 as "writer", pass 2
    // as "reader"; a single matrix beyond the cache keeps the streamed kernels and one launch per pass
    const size_t matrix_bytes = static_cast<size_t>(n0) * static_cast<size_t>(n1) * elem_bytes();
    // (random data, tools/perf_cache.py: 1024^2 x 16 / 32 / 64 / 256 +6 / +13 / +7 / +2 %, 512^2 x 128 +9 %; a 32 MiB
    //  batch is 18 % faster with the streamed kernels, hence the lower bound)
    // SPLIT_COMPLEX storage: the same chunks and policies where both passes have a registered twin with its split form
    // (round 6; rows of 256 ... 2048 points over a column pass of 64 ... 256 points: C5 in split storage ran the streamed kernels in
    // one launch per pass at 0.34 against the interleaved plan's 0.39 -- profiles/r6_notes.md section 8); else streamed
    bool split_twins = false;
    if (split && !kn.no_split_2d_cached) {
      const rows2d_kernel* w = find_rows2d_registered(n1, n0, 1, true);
      if (w != nullptr) {
        const strided_kernel* r = find_strided(n0 / w->rc, true, false, static_cast<long long>(w->rc) * n1, 2, false);
        split_twins = r != nullptr && r->launch_split != nullptr;
      }
    }
    const bool cached = (!split || split_twins) && cache_chunk_bytes() >= matrix_bytes &&
                        matrix_bytes * static_cast<size_t>(nmat) >= cache_chunk_bytes() / 2;
    const long long chunk_mats = cached ? even_chunks(std::max<long long>(1, std::min<long long>(
                                                          nmat, static_cast<long long>(cache_chunk_bytes() / matrix_bytes))),
                                                      nmat)
                                        : nmat;
    const rows2d_kernel* rk = find_rows2d(n1, n0, cached ? 1 : 0, split);
    const bool range_ok = static_cast<unsigned long long>(n0) * static_cast<unsigned long long>(n1) * elem_bytes() <
                          0xFFFFFFF0ull;
    const size_t all_bytes = static_cast<size_t>(B) * static_cast<size_t>(total) * elem_bytes();
    const bool alias_ok = desc.placement != PFFT_IN_PLACE || all_bytes <= global_chunk_bytes();
    if (rk != nullptr && range_ok && alias_ok) {
      const long long m = n0 / rk->rc;
      const long long cols = static_cast<long long>(rk->rc) * n1;
      addressing a{static_cast<long long>(vout.offset), cols, 1, n0 * n1};
      std::vector<stage> tail;
      pfft_dim_info_t di{};
      tail_policy = cached ? 2 : 0;
      // the attempt is planned into temporaries: when it is rejected below, whatever plan_1d reserved for it (scratch
      // of a two-stage column plan, chunk groups) is given back (its small twiddle tables stay uploaded
      // until the plan goes away)
      const size_t saved_scratch = scratch_bytes, saved_half = overlap_scratch_half;
      const int saved_groups = n_chunk_groups;
      const int tier = plan_1d(tail, m, nmat * cols, cols, BUF_OUT, a, BUF_OUT, a, false, scale, backward, &di);
      tail_policy = 0;
      const bool accepted = tier == PFFT_TIER_WORKGROUP && tail.size() == 1 && tail[0].strided != nullptr;
      if (!accepted) {
        scratch_bytes = saved_scratch;
        overlap_scratch_half = saved_half;
        n_chunk_groups = saved_groups;
      }
      if (accepted) {
        st.push_back(make_rows2d_stage(rk, nmat, n0, static_cast<long long>(vin.offset),
                                       static_cast<long long>(vout.offset), backward));
        tail[0].alias_scratch = 2;
        if (chunk_mats < nmat) {  // pass 1 and pass 2 advance together, chunk_mats matrices at a time
          const int group_id = n_chunk_groups++;
          stage& s1 = st.back();
          s1.chunk_group = group_id;
          s1.chunk_batches = chunk_mats;
          s1.ffts_per_batch = n0;
          s1.in_batch_dist = s1.out_batch_dist = n0 * n1;
          tail[0].chunk_group = group_id;
          tail[0].chunk_batches = chunk_mats;
          tail[0].ffts_per_batch = cols;
          tail[0].in_batch_dist = tail[0].out_batch_dist = n0 * n1;
          regrid_for_chunk(s1, chunk_mats * n0);
          regrid_for_chunk(tail[0], chunk_mats * cols);
        }
        st.push_back(tail[0]);
        if (desc.placement == PFFT_IN_PLACE) {
          alias_scratch_bytes = std::max(alias_scratch_bytes, static_cast<size_t>(chunk_mats) * matrix_bytes);
        }
        two_pass_chunk_bytes = std::max(two_pass_chunk_bytes, static_cast<size_t>(chunk_mats) * matrix_bytes);
        if (record) {
          pfft_dim_info_t& d1 = info.dims[rank - 1];
          d1.length = static_cast<uint64_t>(n1);
          d1.tier = PFFT_TIER_WORKGROUP;
          d1.n_factors = rk->n_radices;
          for (int i = 0; i < rk->n_radices; ++i) d1.factors[i] = rk->radices[i];
          d1.workgroup_size = rk->wg;
          d1.ffts_per_workgroup = rk->rc;
          d1.lds_bytes = rk->lds_bytes;
          pfft_dim_info_t& d0 = info.dims[rank - 2];
          d0 = di;  // the column dimension: radix RC (fused into pass 1), then the factors of n0 / RC
          d0.length = static_cast<uint64_t>(n0);
          const int nf = std::min<int>(di.n_factors, PFFT_MAX_FACTORS - 1);
          d0.factors[0] = rk->rc;
          for (int i = 0; i < nf; ++i) d0.factors[i + 1] = di.factors[i];
          d0.n_factors = nf + 1;
        }
        fused_from = rank - 2;  // `inner` is accumulated over lengths[fused_from..rank) below
      }
    }
  }
  if (fused_from == rank) {
    const long long last = static_cast<long long>(desc.lengths[rank - 1]);
    addressing ia{static_cast<long long>(vin.offset), 1, last, 0};
    addressing oa{static_cast<long long>(vout.offset), 1, last, 0};
    const long long count = B * (total / last);
    plan_1d(st, last, count, count, BUF_IN, ia, BUF_OUT, oa, true, scale, backward,
            record ? &info.dims[rank - 1] : nullptr);
    inner = last;
    fused_from = rank - 1;
  } else {
    for (int i = fused_from; i < rank; ++i) inner *= static_cast<long long>(desc.lengths[i]);
  }
  for (int i = fused_from - 1; i >= 0; --i) {
    const long long n = static_cast<long long>(desc.lengths[i]);
    const long long outer_count = B * (total / (inner * n));
    addressing a{static_cast<long long>(vout.offset), inner, 1, inner * n};
    plan_1d(st, n, outer_count * inner, inner, BUF_OUT, a, BUF_OUT, a, false, 1.0, backward,
            record ? &info.dims[i] : nullptr);
    inner *= n;
  }
}

}  // namespace pfa
