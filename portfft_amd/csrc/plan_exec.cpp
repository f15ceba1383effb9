// plan_t: execute -- binding the user's pointers, chunked and overlapped launches -- and the C ABI on top of it
// (include/portfft_amd.h).
#include "plan.hpp"

#include <cstdio>

namespace pfa {

/// intermediate of the two-pass 2-D plan when the caller's buffers alias: allocated at commit for IN_PLACE
/// descriptors, on first use when an OUT_OF_PLACE plan is executed with in == out
/// (the lazy path is serialised, refuses to allocate inside a stream capture -- hipMalloc is not capturable: run the
///  aliasing execute once before capturing, or commit the descriptor IN_PLACE)
void plan_t::ensure_alias_scratch() {
  if (alias_scratch != nullptr) return;
  static std::mutex m;
  std::lock_guard<std::mutex> lock(m);
  if (alias_scratch != nullptr) return;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) {
    fail(PFFT_INVALID_CONFIGURATION,
         "an OUT_OF_PLACE plan executed with aliasing buffers needs an intermediate that cannot be allocated during "
         "stream capture: execute it once before capturing, or commit the descriptor IN_PLACE");
  }
  if (alias_scratch_bytes == 0) alias_scratch_bytes = two_pass_chunk_bytes;
  hip_check(hipMalloc(&alias_scratch, alias_scratch_bytes), "hipMalloc(2-D intermediate)");
}

/// run stage `s` for the user transforms [b0, b0 + nb) (chunked stages) or entirely (nb < 0)
void plan_t::run_stage(const stage& s, const void* in_re, const void* in_im, void* out_re, void* out_im, long long b0,
                       long long nb, const launch_ctx& lc) {
  const bool on_aux = lc.on_aux, any_order = lc.any_order;
  const size_t scratch_shift = lc.scratch_shift;
  const long long in_shift = nb < 0 ? 0 : b0 * s.in_batch_dist;    // elements
  const long long out_shift = nb < 0 ? 0 : b0 * s.out_batch_dist;  // elements
  const long long count = nb < 0 ? s.count : nb * s.ffts_per_batch;
  const size_t sb = static_cast<size_t>(scalar_bytes());
  const bool split = desc.complex_storage == PFFT_SPLIT_COMPLEX;
  hipStream_t stream = on_aux ? aux_stream : this->stream;
  char* const scratch = this->scratch == nullptr ? nullptr : static_cast<char*>(this->scratch) + scratch_shift;
  // resolve buffers: user buffers follow the descriptor's storage, scratch is always interleaved
  auto base_re = [&](int buf, bool is_in) -> const char* {
    if (buf == BUF_SCRATCH) return static_cast<const char*>(scratch);
    if (buf == BUF_IN) return static_cast<const char*>(in_re);
    (void)is_in;
    return static_cast<const char*>(out_re);
  };
  auto base_im = [&](int buf) -> const char* {
    if (buf == BUF_SCRATCH) return static_cast<const char*>(scratch) + sb;
    if (buf == BUF_IN) return split ? static_cast<const char*>(in_im) : static_cast<const char*>(in_re) + sb;
    return split ? static_cast<const char*>(out_im) : static_cast<const char*>(out_re) + sb;
  };
  auto step_of = [&](int buf) { return (buf == BUF_SCRATCH || !split) ? 2 : 1; };
  // (split storage: either pair of planes aliasing means the pass cannot write its output over its input)
  const bool aliased = s.alias_scratch != 0 && (in_re == out_re || (split && in_im != nullptr && in_im == out_im));
  if (aliased) ensure_alias_scratch();
  if (s.xcd != nullptr) {  // one launch for the whole batch: stage A and stage B tasks from per-XCD queues
    xcd_args x = s.xa;
    x.a.in = static_cast<const char*>(in_re) + static_cast<size_t>(s.in_addr.offset) * elem_bytes();
    x.a.out = scratch;
    x.b.in = scratch;
    x.b.out = static_cast<char*>(out_re) + static_cast<size_t>(s.out_addr.offset) * elem_bytes();
    x.ctl = static_cast<unsigned*>(xcd_ctl);
    x.tmap = static_cast<unsigned*>(xcd_tmap);
    x.report = xcd_report;
    // The persistent launch, then its recovery launch: every work-group of the latter reads one word and leaves unless a
    // hand-off wait of the former gave up, in which case it recomputes what is missing IN STREAM ORDER -- the
    // submission's event (the stop event of the last launch) and everything queued behind the execute see valid data.
    // Aliasing buffers: stage B from the slot rings first (the input of those transforms is already overwritten).
    // A launch that gave up is recomputed silently in stream order -- but it first stalled the stream for seconds (2^24
    // polls of the waiting wave, stockham_xcd.hpp XCD_SPIN_LIMIT).  The report word lives in pinned host memory: look at it
    // WITHOUT synchronising, here, on the next execute, and say so once (latency spikes without a diagnostic).
    if (!xcd_recovery_warned && xcd_report != nullptr && __atomic_load_n(xcd_report, __ATOMIC_RELAXED) != 0) {
      xcd_recovery_warned = true;
      std::fprintf(stderr,
                   "[portfft_amd] an XCD-local four-step launch of this plan gave up a hand-off wait and was recomputed by its "
                   "recovery launch (%u so far; the results are valid, the execute stalled for up to ~3 s). "
                   "pfft_plan_info_t::xcd_recoveries counts them, PFFT_XCD_CHECK=1 turns them into errors, "
                   "PFFT_NO_XCD_LOCAL=1 keeps the two-launch plan.\n",
                   __atomic_load_n(xcd_report, __ATOMIC_RELAXED));
    }
    const hipEvent_t stop = take_stop_event();
    hip_check(s.xcd->launch(stream, s.grid, s.lds_bytes, x, s.backward), "kernel launch");
    const bool aliasing = in_re == out_re;
    if (aliasing) {
      hip_check(s.xcd->launch_recover(stream, s.recover_grid, s.lds_bytes, x, s.backward, XCD_RECOVER_STAGE_B), "kernel launch");
    }
    if (stop != nullptr) arm_stop_event(stop);
    hip_check(s.xcd->launch_recover(stream, s.recover_grid, s.lds_bytes, x, s.backward,
                                    aliasing ? XCD_RECOVER_REST : XCD_RECOVER_ALL),
              "kernel launch");
    if (kn.xcd_check) check_xcd_recoveries();
    return;
  }
  if (s.rows2d != nullptr) {
    rows2d_args a = s.ra;
    a.any_order = any_order ? 1 : 0;
    if (nb >= 0) a.nmat = nb;  // chunked: matrices [b0, b0 + nb)
    const size_t unit = split ? sb : elem_bytes();
    const size_t io = static_cast<size_t>(s.in_offset + in_shift) * unit;
    const size_t oo = static_cast<size_t>(s.out_offset + out_shift) * unit;
    a.in = static_cast<const char*>(in_re) + io;
    a.out = aliased ? static_cast<char*>(alias_scratch)  // one chunk at a time goes through the scratch
                    : static_cast<char*>(out_re) + oo;
    if (split) {  // the scratch of an in-place execute holds the two planes one after the other
      a.in_im = static_cast<const char*>(in_im) + io;
      a.out_im = aliased ? static_cast<char*>(alias_scratch) + alias_scratch_bytes / 2 : static_cast<char*>(out_im) + oo;
    }
    const long long groups = a.nmat * (a.n0 / s.rows2d->rc);
    unsigned grid = static_cast<unsigned>(std::min<long long>(s.grid, std::max<long long>(groups, 1)));
    hip_check(s.rows2d->launch != nullptr
                  ? (split ? s.rows2d->launch_split : s.rows2d->launch)(stream, grid, a, s.backward)
                  : jit_launch_rows2d(s.rows2d, stream, grid, a, s.backward),
              "kernel launch");
    return;
  }
  if (s.strided != nullptr) {
    strided_args a = s.sa;
    a.any_order = any_order ? 1 : 0;
    a.total = count;
    const long long groups = strided_groups(count, a.inner, s.strided->fpw);
    unsigned grid = static_cast<unsigned>(std::min<long long>(s.grid, std::max<long long>(groups, 1)));
    // column-shaped input in segments narrower than a 128-byte line (fp32 n = 2048 stages: 8 columns; the planes of
    // SPLIT_COMPLEX data at 16 fp32 / 8 fp64 columns): neighbouring groups on one XCD (strided_args::pair_xcd).
    // Only the plain strided kernel forms honour it; the grid becomes a multiple of 16.
    {
      const bool in_user_split = split && s.in_buf != BUF_SCRATCH;
      const size_t seg = static_cast<size_t>(s.strided->fpw) * (in_user_split ? sb : elem_bytes());
      const bool column_in = a.in_fdist == 1 && a.in_gdist == 0 && a.in_stride > 1;
      if (column_in && seg < 128 && s.row_mode == 0 && s.tiled_in == 0 && s.strided->fpw > 1 && grid >= 32 &&
          groups >= 32 && kn.pair_xcd) {
        a.pair_xcd = 1;
        grid &= ~15u;
      }
      // ... and a row pitch that is no multiple of a line (68640 = 104 x 660: 5280-byte rows; the columns of a 1000 x 1000
      // matrix): every segment straddles one line more than it fills and shares it with the neighbour group -- the blocks
      // of an XCD walk neighbouring groups (strided_group_walk, pair_xcd 2) so that the shared line is fetched once
      const size_t pitch = static_cast<size_t>(a.in_stride) * (in_user_split ? sb : elem_bytes());
      // ... or a column-shaped OUTPUT at such a pitch (the four-step stage B of 68640 = 104 x 660 writes 128-byte segments at a
      // pitch of 832 bytes): the partial lines of two neighbouring groups meet in one L2
      const bool out_user_split = split && s.out_buf != BUF_SCRATCH;
      const bool column_out = a.out_fdist == 1 && a.out_gdist == 0 && a.out_stride > 1 && a.out_tile_shift == 0;
      const size_t opitch = static_cast<size_t>(a.out_stride) * (out_user_split ? sb : elem_bytes());
      // (kernels compiled at commit only: the pre-compiled instantiations keep the loop of rounds 1-5)
      // (the row-staged forms of such entries walk the same way)
      if (a.pair_xcd == 0 && ((column_in && a.in_tile_shift == 0 && pitch % 128 != 0) || (column_out && opitch % 128 != 0)) &&
          s.tiled_in == 0 && s.strided->launch == nullptr && s.strided->fpw > 1 && grid >= 64 && kn.xcd_contig) {
        a.pair_xcd = 2;
      }
    }
    if (split && (s.in_buf == BUF_SCRATCH) != (s.out_buf == BUF_SCRATCH)) {  // mixed storage (four-step stages)
      const bool in_user = s.in_buf != BUF_SCRATCH;
      const size_t iu = in_user ? sb : elem_bytes(), ou = in_user ? elem_bytes() : sb;
      const size_t io = static_cast<size_t>(s.in_addr.offset + in_shift) * iu;
      const size_t oo = static_cast<size_t>(s.out_addr.offset + out_shift) * ou;
      a.in = base_re(s.in_buf, true) + io;
      a.in_im = in_user ? base_im(s.in_buf) + io : nullptr;
      a.out = const_cast<char*>(base_re(s.out_buf, false)) + oo;
      a.out_im = in_user ? nullptr : const_cast<char*>(base_im(s.out_buf)) + oo;
      if (s.tiled_in != 0 && !in_user) {
        hip_check(jit_launch_strided_mixed_tin(s.strided, stream, grid, a, s.backward), "kernel launch");
        return;
      }
      hip_check(s.row_mode == 1 && !in_user ? jit_launch_strided_row_mixed(s.strided, stream, grid, a, s.backward)
                                            : jit_launch_strided_mixed(s.strided, stream, grid, a, s.backward, in_user ? 2 : 3),
                "kernel launch");
      return;
    }
    if (split && s.in_buf != BUF_SCRATCH) {  // both sides are user buffers
      const size_t io = static_cast<size_t>(s.in_addr.offset + in_shift) * sb;
      const size_t oo = static_cast<size_t>(s.out_addr.offset + out_shift) * sb;
      a.in = base_re(s.in_buf, true) + io;
      a.in_im = base_im(s.in_buf) + io;
      if (aliased && s.alias_scratch == 2) {  // two-pass 2-D plan, in-place execute: pass 1 left planes in the scratch
        a.in = alias_scratch;
        a.in_im = static_cast<const char*>(alias_scratch) + alias_scratch_bytes / 2;
      }
      a.out = const_cast<char*>(base_re(s.out_buf, false)) + oo;
      a.out_im = const_cast<char*>(base_im(s.out_buf)) + oo;
      hip_check(s.strided->launch != nullptr ? s.strided->launch_split(stream, grid, a, s.backward)
                                             : jit_launch_strided_split(s.strided, stream, grid, a, s.backward, s.store_modifier),
                "kernel launch");
      return;
    }
    a.in = base_re(s.in_buf, true) + static_cast<size_t>(s.in_addr.offset + in_shift) * elem_bytes();
    if (aliased && s.alias_scratch == 2) a.in = alias_scratch;  // two-pass 2-D plan, in-place execute
    a.out = const_cast<char*>(base_re(s.out_buf, false)) +
            static_cast<size_t>(s.out_addr.offset + out_shift) * elem_bytes();
    if (s.row_mode != 0 && s.store_modifier == 0) {
      hip_check(s.strided->launch_row != nullptr
                    ? s.strided->launch_row(stream, grid, a, s.backward, s.row_mode - 1)
                    : jit_launch_strided_row(s.strided, stream, grid, a, s.backward, s.row_mode - 1),
                "kernel launch");
      return;
    }
    if (s.tiled_in != 0) {
      hip_check(s.tiled_in == 2 ? s.strided->launch_tin_w(stream, grid, a, s.backward)
                                : s.strided->launch_tin(stream, grid, a, s.backward),
                "kernel launch");
      return;
    }
    hip_check(s.strided->launch != nullptr
                  ? s.strided->launch(stream, grid, a, s.backward, s.store_modifier)
                  : jit_launch_strided(s.strided, stream, grid, a, s.backward, s.store_modifier),
              "kernel launch");
    return;
  }
  if (!s.generic && s.unpacked != nullptr) {
    const bool user_split = split && s.in_buf != BUF_SCRATCH;
    const size_t unit = user_split ? sb : elem_bytes();
    const char* i_re = base_re(s.in_buf, true) + static_cast<size_t>(s.in_offset) * unit;
    const char* i_im = base_im(s.in_buf) + static_cast<size_t>(s.in_offset) * unit;
    char* o_re = const_cast<char*>(base_re(s.out_buf, false)) + static_cast<size_t>(s.out_offset) * unit;
    char* o_im = const_cast<char*>(base_im(s.out_buf)) + static_cast<size_t>(s.out_offset) * unit;
    hip_check(jit_launch_unpacked(s.unpacked, user_split, stream, s.grid, i_re, i_im, o_re, o_im, s.tw, s.count,
                                  s.scale, s.backward, static_cast<unsigned>(s.in_addr.stride),
                                  static_cast<unsigned>(s.in_addr.dist_inner),
                                  static_cast<unsigned>(s.out_addr.stride),
                                  static_cast<unsigned>(s.out_addr.dist_inner)),
              "kernel launch");
    return;
  }
  if (!s.generic) {
    if (split) {  // spec stages only touch user buffers when the storage is split (plan_1d)
      const size_t io = static_cast<size_t>(s.in_offset) * sb, oo = static_cast<size_t>(s.out_offset) * sb;
      const char* sr = static_cast<const char*>(s.in_buf == BUF_IN ? in_re : out_re);
      const char* si = static_cast<const char*>(s.in_buf == BUF_IN ? in_im : out_im);
      auto launch_split = s.spec->launch != nullptr ? s.spec->launch_split : nullptr;
      hip_check(launch_split != nullptr
                    ? launch_split(stream, s.grid, sr + io, si + io, static_cast<char*>(out_re) + oo,
                                   static_cast<char*>(out_im) + oo, s.tw, s.count, s.scale, s.backward)
                    : jit_launch_spec_split(s.spec, stream, s.grid, sr + io, si + io, static_cast<char*>(out_re) + oo,
                                            static_cast<char*>(out_im) + oo, s.tw, s.count, s.scale, s.backward),
                "kernel launch");
      return;
    }
    const char* i = base_re(s.in_buf, true) + static_cast<size_t>(s.in_offset) * elem_bytes();
    char* o = const_cast<char*>(base_re(s.out_buf, false)) + static_cast<size_t>(s.out_offset) * elem_bytes();
    hip_check(s.spec->launch != nullptr
                  ? s.spec->launch(stream, s.grid, i, o, s.tw, s.count, s.scale, s.backward)
                  : jit_launch_spec(s.spec, stream, s.grid, i, o, s.tw, s.count, s.scale, s.backward),
              "kernel launch");
    return;
  }
  generic_args g = s.ga;
  g.total_count = count;
  g.in_step = step_of(s.in_buf);
  g.out_step = step_of(s.out_buf);
  const size_t ioff = static_cast<size_t>(s.in_addr.offset + in_shift);
  const size_t ooff = static_cast<size_t>(s.out_addr.offset + out_shift);
  g.in_re = base_re(s.in_buf, true) + ioff * sb * static_cast<size_t>(g.in_step);
  g.in_im = base_im(s.in_buf) + ioff * sb * static_cast<size_t>(g.in_step);
  g.out_re = const_cast<char*>(base_re(s.out_buf, false)) + ooff * sb * static_cast<size_t>(g.out_step);
  g.out_im = const_cast<char*>(base_im(s.out_buf)) + ooff * sb * static_cast<size_t>(g.out_step);
  const hipError_t e = desc.precision == PFFT_PRECISION_F64 ? launch_generic_f64(stream, s.grid, s.lds_bytes, g)
                                                            : launch_generic_f32(stream, s.grid, s.lds_bytes, g);
  hip_check(e, "kernel launch");
}

/// a two-launch chunk group with several chunks whose chunks do not share an intermediate buffer
bool plan_t::overlappable(const std::vector<stage>& st, size_t i, size_t j, bool several_chunks, bool aliased) const {
  if (!chunk_overlap_enabled() || j - i != 2 || !several_chunks) return false;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) {
    return false;  // a captured execute is a plain chain of kernel nodes
  }
  const stage& a = st[i];
  const stage& b = st[i + 1];
  if (aliased && (a.alias_scratch != 0 || b.alias_scratch != 0)) return false;  // one scratch chunk for all chunks
  const bool uses_scratch = a.in_buf == BUF_SCRATCH || a.out_buf == BUF_SCRATCH || b.in_buf == BUF_SCRATCH ||
                            b.out_buf == BUF_SCRATCH;
  return !uses_scratch || overlap_scratch_half > 0;
}

/// chunk c: first launch on the plan's stream, second launch on aux_stream behind an event; the plan's stream joins
/// at the end.  The scratch (four-step plans) alternates between its two halves.
void plan_t::run_chunks_overlapped(const stage& a, const stage& b, long long batches, long long chunk_batches,
                                   const void* in_re, const void* in_im, void* out_re, void* out_im) {
  if (overlap_mode == 2) {
    // Every second launch keeps its barrier, so when the barrier-free first launch of chunk c starts, only the
    // second launch of chunk c - 1 can still be running: other matrices, the other half of the scratch.
    size_t c = 0;
    for (long long b0 = 0; b0 < batches; b0 += chunk_batches, ++c) {
      const long long nb = std::min(chunk_batches, batches - b0);
      launch_ctx lc;
      lc.scratch_shift = (c & 1) * overlap_scratch_half;
      lc.any_order = c > 0;
      run_stage(a, in_re, in_im, out_re, out_im, b0, nb, lc);
      lc.any_order = false;
      run_stage(b, in_re, in_im, out_re, out_im, b0, nb, lc);
    }
    return;
  }
  if (aux_stream == nullptr) {
    hip_check(hipStreamCreateWithFlags(&aux_stream, hipStreamNonBlocking), "hipStreamCreate");
  }
  const size_t n_chunks = static_cast<size_t>((batches + chunk_batches - 1) / chunk_batches);
  while (chunk_events.size() < 2 * n_chunks) {  // [c]: first launch of chunk c done, [n_chunks + c]: second launch
    hipEvent_t ev = nullptr;
    hip_check(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
    chunk_events.push_back(ev);
  }
  size_t c = 0;
  for (long long b0 = 0; b0 < batches; b0 += chunk_batches, ++c) {
    const long long nb = std::min(chunk_batches, batches - b0);
    launch_ctx lc;
    lc.scratch_shift = (c & 1) * overlap_scratch_half;
    if (c >= 2 && overlap_scratch_half > 0) {  // this half of the scratch was last read by chunk c - 2
      hip_check(hipStreamWaitEvent(stream, chunk_events[n_chunks + c - 2], 0), "hipStreamWaitEvent");
    }
    run_stage(a, in_re, in_im, out_re, out_im, b0, nb, lc);
    hip_check(hipEventRecord(chunk_events[c], stream), "hipEventRecord");
    hip_check(hipStreamWaitEvent(aux_stream, chunk_events[c], 0), "hipStreamWaitEvent");
    lc.on_aux = true;
    run_stage(b, in_re, in_im, out_re, out_im, b0, nb, lc);
    if (c + 1 == n_chunks || overlap_scratch_half > 0) {
      hip_check(hipEventRecord(chunk_events[n_chunks + c], aux_stream), "hipEventRecord");
    }
  }
  hip_check(hipStreamWaitEvent(stream, chunk_events[2 * n_chunks - 1], 0), "hipStreamWaitEvent");
}

/// `completion`: the submission's completion event.  Returns true when it rode on the last launch as that dispatch's
/// stop event (kernels.hpp: arm_stop_event); otherwise the caller records it behind the launches.
bool plan_t::execute(int direction, const void* in_re, const void* in_im, void* out_re, void* out_im,
                     hipEvent_t completion) {
  if (direction != PFFT_FORWARD && direction != PFFT_BACKWARD) {
    fail(PFFT_INVALID_CONFIGURATION, "Invalid direction ", direction);
  }
  if (in_re == nullptr || out_re == nullptr) fail(PFFT_INVALID_CONFIGURATION, "null data pointer");
  device_guard dg(device);  // launches go to the device the plan was committed on, whatever is current
  const std::vector<stage>& st = stages[direction];
  bool rode = false;
  if (completion != nullptr && !st.empty() && st.back().chunk_group < 0 && kn.stop_event_on_launch) {
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) {
      completion = nullptr;  // a captured execute records its event as a node of its own
    }
  } else {
    completion = nullptr;
  }
  for (size_t i = 0; i < st.size();) {
    if (st[i].chunk_group < 0) {
      if (completion != nullptr && i + 1 == st.size()) {
        arm_stop_event(completion);
        struct disarm {  // also when the launch throws
          bool* rode;
          ~disarm() { *rode = take_stop_event() == nullptr; }
        } guard{&rode};
        run_stage(st[i], in_re, in_im, out_re, out_im);
      } else {
        run_stage(st[i], in_re, in_im, out_re, out_im);
      }
      ++i;
      continue;
    }
    size_t j = i;
    while (j < st.size() && st[j].chunk_group == st[i].chunk_group) ++j;
    const long long batches = st[i].count / st[i].ffts_per_batch;
    const long long chunk_batches = std::max<long long>(1, st[i].chunk_batches);
    if (overlappable(st, i, j, batches > chunk_batches, in_re == out_re || (in_im != nullptr && in_im == out_im))) {
      run_chunks_overlapped(st[i], st[i + 1], batches, chunk_batches, in_re, in_im, out_re, out_im);
      i = j;
      continue;
    }
    for (long long b0 = 0; b0 < batches; b0 += chunk_batches) {
      const long long nb = std::min(chunk_batches, batches - b0);
      for (size_t k = i; k < j; ++k) run_stage(st[k], in_re, in_im, out_re, out_im, b0, nb);
    }
    i = j;
  }
  return rode;
}

void plan_t::check_xcd_recoveries() {
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) return;
  hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
  const unsigned n = __atomic_load_n(xcd_report, __ATOMIC_RELAXED);
  if (n != xcd_recoveries_seen) {
    xcd_recoveries_seen = n;
    fail(PFFT_INTERNAL_ERROR, "XCD-local four-step launch: ", xcd_report[1], " hand-off waits gave up (first: site ",
         xcd_report[2], ", local transform ", xcd_report[3], ", wanted ", xcd_report[4], ", saw ", xcd_report[5],
         "); the execute was recomputed by its recovery launch (PFFT_XCD_CHECK=1 reports this as an error)");
  }
}

}  // namespace pfa


struct pfft_plan_t {
  std::unique_ptr<pfa::plan_t> impl;
};

namespace {
/// Completion events are recycled: creating and destroying a hipEvent_t per submission costs more than the launch
/// of a small transform.  pfft_event_destroy returns the event to this pool (per device), new events come from it.
struct event_pool {
  std::mutex m;
  std::vector<std::pair<int, hipEvent_t>> free_list;  // (device, event)
  hipEvent_t get(int device) {
    {
      std::lock_guard<std::mutex> lock(m);
      for (size_t i = free_list.size(); i-- > 0;) {
        if (free_list[i].first == device) {
          hipEvent_t ev = free_list[i].second;
          free_list[i] = free_list.back();
          free_list.pop_back();
          return ev;
        }
      }
    }
    hipEvent_t ev = nullptr;
    const hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipEventCreate: ", hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(m);
    owner[ev] = device;
    return ev;
  }
  /// true when the event was taken back (events not created here are destroyed by the caller)
  bool put(hipEvent_t ev) {
    std::lock_guard<std::mutex> lock(m);
    const auto it = owner.find(ev);
    if (it == owner.end()) return false;
    if (free_list.size() >= 1024) {
      owner.erase(it);
      return false;
    }
    free_list.emplace_back(it->second, ev);
    return true;
  }
  std::map<hipEvent_t, int> owner;
};
event_pool& events() {
  static event_pool* p = new event_pool();  // never destroyed: events may outlive static destruction order
  return *p;
}

/// dependencies in, completion event out (shared by the two _ex entry points)
template <typename Run>
void execute_with_events(pfft_plan_t* plan, int32_t n_deps, void* const* deps, void** event_out, Run&& run) {
  pfa::plan_t& p = *plan->impl;
  pfa::device_guard dg(p.device);
  if (n_deps < 0 || (n_deps > 0 && deps == nullptr)) pfa::fail(PFFT_INVALID_CONFIGURATION, "invalid dependency list");
  for (int32_t i = 0; i < n_deps; ++i) {
    if (deps[i] == nullptr) continue;  // a default-constructed event: nothing to wait for
    const hipError_t e = hipStreamWaitEvent(p.stream, static_cast<hipEvent_t>(deps[i]), 0);
    if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipStreamWaitEvent: ", hipGetErrorString(e));
  }
  if (event_out == nullptr) {
    (void)run(nullptr);
    return;
  }
  *event_out = nullptr;
  hipEvent_t ev = events().get(p.device);
  bool rode = false;
  try {
    rode = run(ev);  // true: the event is the stop event of the last dispatch, no packet of its own
  } catch (...) {
    (void)events().put(ev);
    throw;
  }
  if (!rode) {
    const hipError_t e = hipEventRecord(ev, p.stream);
    if (e != hipSuccess) {
      (void)events().put(ev);
      pfa::fail(PFFT_HIP_ERROR, "hipEventRecord: ", hipGetErrorString(e));
    }
  }
  *event_out = ev;
}
}  // namespace

extern "C" {

pfft_status pfft_plan_create(const pfft_desc_t* desc, void* hip_stream, pfft_plan_t** plan) {
  return pfa::guarded([&] {
    if (desc == nullptr || plan == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null argument");
    *plan = nullptr;
    auto p = std::make_unique<pfft_plan_t>();
    p->impl = std::make_unique<pfa::plan_t>(*desc, static_cast<hipStream_t>(hip_stream));
    *plan = p.release();
  });
}

pfft_status pfft_plan_destroy(pfft_plan_t* plan) {
  return pfa::guarded([&] { delete plan; });
}

pfft_status pfft_plan_get_info(const pfft_plan_t* plan, pfft_plan_info_t* info) {
  return pfa::guarded([&] {
    if (plan == nullptr || info == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null argument");
    *info = plan->impl->info;
    if (plan->impl->xcd_report != nullptr) info->xcd_recoveries = __atomic_load_n(plan->impl->xcd_report, __ATOMIC_RELAXED);
  });
}

pfft_status pfft_execute(pfft_plan_t* plan, int32_t direction, const void* in, void* out) {
  return pfa::guarded([&] {
    if (plan == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null plan");
    if (plan->impl->desc.complex_storage != PFFT_INTERLEAVED_COMPLEX) {
      // committed_descriptor_impl.hpp:862-871
      pfa::fail(PFFT_INVALID_CONFIGURATION,
                "To use interleaved data layout, the descriptor.complex_storage must be INTERLEAVED_COMPLEX");
    }
    plan->impl->execute(direction, in, nullptr, out, nullptr);
  });
}

pfft_status pfft_execute_split(pfft_plan_t* plan, int32_t direction, const void* in_real, const void* in_imag,
                               void* out_real, void* out_imag) {
  return pfa::guarded([&] {
    if (plan == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null plan");
    if (plan->impl->desc.complex_storage != PFFT_SPLIT_COMPLEX) {
      pfa::fail(PFFT_INVALID_CONFIGURATION,
                "To use split data layout, the descriptor.complex_storage must be SPLIT_COMPLEX");
    }
    if (in_imag == nullptr || out_imag == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null imaginary pointer");
    plan->impl->execute(direction, in_real, in_imag, out_real, out_imag);
  });
}

pfft_status pfft_execute_ex(pfft_plan_t* plan, int32_t direction, const void* in, void* out, int32_t n_deps,
                            void* const* deps, void** event_out) {
  return pfa::guarded([&] {
    if (plan == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null plan");
    if (plan->impl->desc.complex_storage != PFFT_INTERLEAVED_COMPLEX) {
      pfa::fail(PFFT_INVALID_CONFIGURATION,
                "To use interleaved data layout, the descriptor.complex_storage must be INTERLEAVED_COMPLEX");
    }
    execute_with_events(plan, n_deps, deps, event_out,
                        [&](hipEvent_t ev) { return plan->impl->execute(direction, in, nullptr, out, nullptr, ev); });
  });
}

pfft_status pfft_execute_split_ex(pfft_plan_t* plan, int32_t direction, const void* in_real, const void* in_imag,
                                  void* out_real, void* out_imag, int32_t n_deps, void* const* deps,
                                  void** event_out) {
  return pfa::guarded([&] {
    if (plan == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null plan");
    if (plan->impl->desc.complex_storage != PFFT_SPLIT_COMPLEX) {
      pfa::fail(PFFT_INVALID_CONFIGURATION,
                "To use split data layout, the descriptor.complex_storage must be SPLIT_COMPLEX");
    }
    if (in_imag == nullptr || out_imag == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null imaginary pointer");
    execute_with_events(plan, n_deps, deps, event_out,
                        [&](hipEvent_t ev) {
                          return plan->impl->execute(direction, in_real, in_imag, out_real, out_imag, ev);
                        });
  });
}

pfft_status pfft_event_wait(void* event) {
  return pfa::guarded([&] {
    if (event == nullptr) return;
    const hipError_t e = hipEventSynchronize(static_cast<hipEvent_t>(event));
    if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipEventSynchronize: ", hipGetErrorString(e));
  });
}

pfft_status pfft_event_query(void* event, int32_t* done) {
  return pfa::guarded([&] {
    if (done == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null argument");
    *done = 1;
    if (event == nullptr) return;
    const hipError_t e = hipEventQuery(static_cast<hipEvent_t>(event));
    if (e == hipErrorNotReady) {
      *done = 0;
    } else if (e != hipSuccess) {
      pfa::fail(PFFT_HIP_ERROR, "hipEventQuery: ", hipGetErrorString(e));
    }
  });
}

pfft_status pfft_event_destroy(void* event) {
  return pfa::guarded([&] {
    if (event == nullptr) return;
    if (events().put(static_cast<hipEvent_t>(event))) return;  // recycled
    const hipError_t e = hipEventDestroy(static_cast<hipEvent_t>(event));
    if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipEventDestroy: ", hipGetErrorString(e));
  });
}

pfft_status pfft_queue_copy(void* hip_stream, const void* src, void* dst, size_t bytes, int32_t n_deps,
                            void* const* deps, void** event_out) {
  return pfa::guarded([&] {
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (n_deps < 0 || (n_deps > 0 && deps == nullptr)) pfa::fail(PFFT_INVALID_CONFIGURATION, "invalid dependency list");
    for (int32_t i = 0; i < n_deps; ++i) {
      if (deps[i] == nullptr) continue;
      const hipError_t e = hipStreamWaitEvent(st, static_cast<hipEvent_t>(deps[i]), 0);
      if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipStreamWaitEvent: ", hipGetErrorString(e));
    }
    if (bytes > 0) {
      const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st);
      if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipMemcpyAsync: ", hipGetErrorString(e));
    }
    if (event_out != nullptr) {
      *event_out = nullptr;
      int device = 0;
      (void)hipGetDevice(&device);
      hipEvent_t ev = events().get(device);
      const hipError_t e = hipEventRecord(ev, st);
      if (e != hipSuccess) {
        (void)events().put(ev);
        pfa::fail(PFFT_HIP_ERROR, "hipEventRecord: ", hipGetErrorString(e));
      }
      *event_out = ev;
    }
  });
}

pfft_status pfft_queue_wait(void* hip_stream) {
  return pfa::guarded([&] {
    const hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipStreamSynchronize: ", hipGetErrorString(e));
  });
}

pfft_status pfft_plan_clone(const pfft_plan_t* plan, pfft_plan_t** copy) {
  return pfa::guarded([&] {
    if (plan == nullptr || copy == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null argument");
    *copy = nullptr;
    auto p = std::make_unique<pfft_plan_t>();
    p->impl = std::make_unique<pfa::plan_t>(*plan->impl);
    *copy = p.release();
  });
}

pfft_status pfft_plan_wait(pfft_plan_t* plan) {
  return pfa::guarded([&] {
    if (plan == nullptr) pfa::fail(PFFT_INVALID_CONFIGURATION, "null plan");
    const hipError_t e = hipStreamSynchronize(plan->impl->stream);
    if (e != hipSuccess) pfa::fail(PFFT_HIP_ERROR, "hipStreamSynchronize: ", hipGetErrorString(e));
  });
}

}  // extern "C"
