// Error plumbing shared by the host-side sources of libportfft_amd.so.
//
// The reference reports problems as C++ exceptions (/root/reference/src/portfft/common/exceptions.hpp:32-77); inside
// the library we throw pfa::error carrying the matching pfft_status, and the extern "C" layer converts it to a
// status code + thread-local message (no exception crosses the C ABI).
#pragma once
#include <sstream>
#include <stdexcept>
#include <string>

#include "../../include/portfft_amd.h"

namespace pfa {

class error : public std::runtime_error {
 public:
  error(pfft_status st, const std::string& what) : std::runtime_error(what), status(st) {}
  pfft_status status;
};

template <typename... Ts>
[[noreturn]] void fail(pfft_status st, const Ts&... parts) {
  std::stringstream ss;
  (ss << ... << parts);
  throw error(st, ss.str());
}

/// record the message returned by pfft_last_error() on this thread
void set_last_error(const std::string& msg);

/// run `f`, translate exceptions to a status code
template <typename F>
pfft_status guarded(F&& f) {
  try {
    f();
    return PFFT_OK;
  } catch (const error& e) {
    set_last_error(e.what());
    return e.status;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return PFFT_INTERNAL_ERROR;
  } catch (...) {
    set_last_error("unknown exception");
    return PFFT_INTERNAL_ERROR;
  }
}

}  // namespace pfa
