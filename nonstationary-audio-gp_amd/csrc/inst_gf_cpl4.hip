// one group of kernel instantiations of libnagp.so (see nagp_inst.hpp)
#include "nagp_inst.hpp"
NAGP_LIST_GF_CPL4(template __global__)
