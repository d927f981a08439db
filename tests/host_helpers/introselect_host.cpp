// Host build of the PRODUCT's selection routine (csrc/introselect.hpp) so that CPU tests can compare its
// permutation with std::nth_element (through the oracle).
#include "../../ucoslam-cv3_amd/csrc/introselect.hpp"
extern "C" void uh_host_nth_element(uint32_t* v, int n, int nth) { uh_sel::nth_element_desc(v, n, nth); }
extern "C" void uh_host_nth_element_pairing(uint32_t* v, int n, int nth) { uh_sel::nth_element_desc_pairing_host(v, n, nth); }
