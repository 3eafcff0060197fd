// one group of kernel instantiations of libnagp.so (see nagp_inst.hpp)
#include "nagp_inst.hpp"
NAGP_LIST_GF_CPLW(template __global__)
