// error.hpp -- error type carried across the host side; the C ABI turns it into an
// ld_status + ld_last_error() message (the reference panics instead, e.g.
// src/dfire.rs:43,180,247; src/bin/lightdock-rust.rs:172,233).
#pragma once

#include <stdexcept>
#include <string>

#include "lightdock_hip.h"

namespace ld {

class Error : public std::runtime_error {
   public:
    Error(ld_status code, const std::string &what) : std::runtime_error(what), code_(code) {}
    ld_status code() const { return code_; }

   private:
    ld_status code_;
};

}  // namespace ld
