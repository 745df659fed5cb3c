// see include/compat/saena.hpp.  The reference's drivers include this INTERNAL header next to "saena.hpp" (experiments/banded.cpp:1-4);
// the ones that only use the public surface compile with this forwarding header -- the internals themselves (Grid, saena_object,
// saena_matrix members) are not part of the MI355X path's surface.
#pragma once
#include "../saena_mpi.hpp"
