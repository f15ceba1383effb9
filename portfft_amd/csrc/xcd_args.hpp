// Launch-time arguments and control-block layout of the XCD-local four-step kernel (stockham_xcd.hpp).
// Plain C++: shared by the host planner and the kernel.
#pragma once
#include "strided_args.hpp"

namespace pfa {

/// Control block (32-bit words, device memory).  Its counters are zero between launches -- the last work-group to leave
/// a healthy launch clears them again, so a replayed HIP graph needs no memset node:
///   [XCD_W_NEXT]     next unclaimed transform of the launch (queues claim transforms one by one)
///   [XCD_W_EXIT]     work-groups that have left the launch
///   [XCD_W_EPOCH]    launches completed on this block (never cleared; tags the claim-map entries of a launch)
///   [XCD_W_REXIT]    work-groups that have left the recovery launch behind a launch that gave up
///   [XCD_W_TIMEOUT]  hand-off waits of the launch that gave up, then five words describing the first of them.  Zero
///                    after a healthy launch.  A launch that sets it leaves the whole block as it is: the recovery
///                    launches behind it (stockham_xcd_recover_kernel) read the block to find the transforms that are
///                    not complete, recompute them, copy these words to the plan's host report and clear the block.
///   [XCD_W_QUEUES + q * queue_words(...)]  queue q -- one per XCC id:
///       +0                 ticket: tasks handed out
///       +32 .. +32 + 4M    map: M 16-byte entries {launch epoch, local transform + 1, claimed transform + 1 (0: none
///                          left), 0}.  Written with PLAIN stores and read past the L1: writer and readers of a queue
///                          share one XCD, so the entries live in its L2 (an entry read from memory cost every task
///                          ~1.5 us: measured).  Never cleared -- a dirty line of another XCD's L2 could come back
///                          after the clearing stores -- the epoch makes the entries of older launches invalid.
///       +32 + 4M + 64 s    done_a of slot s: stage-A tasks finished, cumulative over the slot's occupants
///       +32 + 4M + 64 s + 32   done_b of slot s: stage-B tasks that have their input in registers
/// Every polled counter sits on a 128-byte line of its own.
enum : unsigned { XCD_W_NEXT = 0, XCD_W_EXIT = 32, XCD_W_EPOCH = 33, XCD_W_REXIT = 34, XCD_W_TIMEOUT = 64, XCD_W_QUEUES = 96 };
constexpr unsigned xcd_queue_words(int slots, int map_log2) {
  return 32u + (4u << map_log2) + 64u * static_cast<unsigned>(slots);
}
constexpr unsigned xcd_ctl_words(int queues, int slots, int map_log2) {
  return XCD_W_QUEUES + static_cast<unsigned>(queues) * xcd_queue_words(slots, map_log2);
}
/// bytes of control words at the end of the kernel's dynamic LDS
constexpr unsigned XCD_LDS_CTL_BYTES = 64;

/// Host report of a plan (pinned host memory, one block per plan copy; written by the last work-group of a recovery
/// launch with system-scope stores): [0] launches that gave up and were recomputed, [1] hand-off waits that gave up in
/// the last of them, [2..6] its first one: site, local transform, wanted, seen, polls.
constexpr unsigned XCD_REPORT_WORDS = 16;

/// Phases of stockham_xcd_recover_kernel (its second argument)
enum : int {
  XCD_RECOVER_STAGE_B = 1,   // stage B again, from the slot rings, for transforms whose stage A is complete (aliasing executes)
  XCD_RECOVER_REST = 2,      // behind phase 1: everything whose stage A is not complete, from the user's input
  XCD_RECOVER_ALL = 3        // alone (input and output do not alias): everything that is not complete, from the input
};

/// One launch = the whole batch.  `a` / `b` are the stage arguments of the two-launch plan with the scratch side
/// rebased: a.out = b.in = the slot rings, a.out_dist_outer = b.in_dist_outer = 0 (the kernel adds the slot's base).
struct xcd_args {
  strided_args a, b;
  unsigned* ctl;
  long long batch;      // transforms of this launch
  int n_queues;         // queues in the control block = XCC ids the device reports (work-groups with another id idle)
  int slots;            // intermediate slots per queue (each one transform); lag < slots
  int map_log2;         // entries of a queue's claim map
  int lag;              // stage-B tickets of a transform come `lag` transforms behind its stage-A tickets
  int lookahead;        // transforms a queue claims ahead of its stage-A tickets
  unsigned max_iters;   // bound of a work-group's ticket loop: (batch + lag + lookahead + 2) * tickets per transform
  unsigned lds_ctl_off; // byte offset of the kernel's XCD_LDS_CTL_BYTES of control words in its dynamic LDS
  unsigned long long* prof;  // tuner builds (PFA_XCD_PROF) only: cycle sums of wave 0 of every work-group
  /// per-transform records, 8 bytes each: {launch epoch + 1, queue << 28 | local transform + 1}, written when a queue
  /// claims the transform -- where the recovery launches find its slot and its hand-off counters (null: tuner builds)
  unsigned* tmap;
  unsigned* report;     // the plan's host report (XCD_REPORT_WORDS words of pinned host memory) or null
};

}  // namespace pfa
