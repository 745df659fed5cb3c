// include/compat/ -- the header NAMES a driver of the reference includes (experiments/Poisson.cpp:1-8: "saena.hpp", "data_struct.h",
// "aux_functions2.h"), each forwarding to include/saena_mpi.hpp: with `-I<repo>/include/compat -I<repo>/include` ahead of the
// reference's include directory such a driver compiles against the MI355X path WITHOUT a source change
// (tests/test_cpp_surface.py does exactly that with the reference's own file where it lies).
#pragma once
#include "../saena_mpi.hpp"
