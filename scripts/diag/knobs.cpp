// knobs.cpp -- linked into the DIAGNOSTIC build of the library only (python -m moss_amd.build --diag -> moss_amd/lib_diag/).
// moss::knob(name, dflt) of moss_amd/csrc/common.h: in the product build a constant, here the integer value of the environment
// variable `name`.  The knobs select kernel variants for A/B timing from scripts/; some give wrong results on purpose.
#include <cstdlib>

namespace moss {

int knob(const char* name, int dflt)
{
    const char* v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : dflt;
}

}  // namespace moss
