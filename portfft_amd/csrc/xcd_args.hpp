// Launch-time arguments and control-block layout of the XCD-local four-step kernel (stockham_xcd.hpp).
// Plain C++: shared by the host planner and the kernel.
#pragma once
#include "strided_args.hpp"

namespace pfa {

/// Control block (32-bit words, device memory, all zero between launches -- the last work-group to leave a launch
/// clears it again, so a replayed HIP graph needs no memset node):
///   [XCD_W_NEXT]     next unclaimed transform of the launch (queues claim transforms one by one)
///   [XCD_W_EXIT]     work-groups that have left the launch
///   [XCD_W_TIMEOUT]  sticky: a bounded spin gave up (never cleared by the kernel; pfft_plan_check reads it)
///   [XCD_W_QUEUES + q * queue_words(...)]  queue q -- one per XCC id:
///       +0                 ticket: tasks handed out
///       +32 .. +32 + 2M    map: M 64-bit entries {tag = local transform + 1, claimed transform + 1 (0: none left)}
///       +32 + 2M + 64 s    done_a of slot s: stage-A tasks finished, cumulative over the slot's occupants
///       +32 + 2M + 64 s + 32   done_b of slot s: stage-B tasks that have their input in registers
/// Every polled word sits on a 128-byte line of its own (the map entries share lines: written once, read by all).
enum : unsigned { XCD_W_NEXT = 0, XCD_W_EXIT = 32, XCD_W_TIMEOUT = 64, XCD_W_QUEUES = 96 };
constexpr unsigned xcd_queue_words(int slots_log2, int map_log2) {
  return 32u + (2u << map_log2) + (64u << slots_log2);
}
constexpr unsigned xcd_ctl_words(int queues, int slots_log2, int map_log2) {
  return XCD_W_QUEUES + static_cast<unsigned>(queues) * xcd_queue_words(slots_log2, map_log2);
}

/// One launch = the whole batch.  `a` / `b` are the stage arguments of the two-launch plan with the scratch side
/// rebased: a.out = b.in = the slot rings, a.out_dist_outer = b.in_dist_outer = 0 (the kernel adds the slot's base).
struct xcd_args {
  strided_args a, b;
  unsigned* ctl;
  long long batch;      // transforms of this launch
  int n_queues;         // queues in the control block = XCC ids the device reports (work-groups with another id idle)
  int slots_log2;       // intermediate slots per queue (each one transform)
  int map_log2;         // entries of a queue's claim map
  int lag;              // stage-B tickets of a transform come `lag` transforms behind its stage-A tickets
  int lookahead;        // transforms a queue claims ahead of its stage-A tickets
  unsigned max_iters;   // bound of a work-group's ticket loop: (batch + lag + lookahead + 2) * tickets per transform
  unsigned lds_ctl_off; // byte offset of the kernel's 16 bytes of control words in its dynamic LDS
  unsigned long long* prof;  // tuner builds (PFA_XCD_PROF) only: cycle sums of wave 0 of every work-group
};

}  // namespace pfa
