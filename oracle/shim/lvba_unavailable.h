// Functions of absent third-party libraries that the reference's translation units mention but that no tested path reaches
// are defined to throw, so that the stand-in library links completely (CPython dlopens with RTLD_NOW) and a wrong turn is
// an error message instead of a crash.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <stdexcept>
#include <string>
[[noreturn]] inline void lvba_unavailable(const char *what)
{
    throw std::runtime_error(std::string("oracle/shim: ") + what + " is not available in this build");
}
