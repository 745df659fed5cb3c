// see include/compat/saena.hpp: the generators and find_split (reference include/aux_functions2.h:12-43) come with the public header
#pragma once
#include "../saena_mpi.hpp"
