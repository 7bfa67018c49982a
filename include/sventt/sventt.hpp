// include/sventt/sventt.hpp -- umbrella header, as in the reference
// (include/sventt/sventt.hpp there): `#include <sventt/sventt.hpp>` brings in
// Modulus, the modmul / layer / transposition / kernel vocabulary, PageMemory
// and NTT<>.  Nothing is gated on __ARM_FEATURE_SVE here: the *SVE* names are
// plan descriptions for the GPU engine (plan_types.hpp).
//
// Build:  g++ -std=c++20 -I<repo>/include app.cpp -L<repo>/sve_ntt_amd -lsventt_hip
#ifndef SVENTT_GPU_HPP_INCLUDED
#define SVENTT_GPU_HPP_INCLUDED

#include "sventt/modulus.hpp"
#include "sventt/utility.hpp"
#include "sventt/vector.hpp"

#include "sventt/plan_types.hpp"

#include "sventt/wrapper.hpp"

#endif /* SVENTT_GPU_HPP_INCLUDED */
