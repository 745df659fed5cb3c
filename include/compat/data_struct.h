// see include/compat/saena.hpp: index_t / nnz_t / value_t (reference include/data_struct.h:36-38) come with the public header
#pragma once
#include "../saena_mpi.hpp"
