// one group of kernel instantiations of libnagp.so (see nagp_inst.hpp)
#include "nagp_inst.hpp"
NAGP_LIST_GAINI(template __global__)
