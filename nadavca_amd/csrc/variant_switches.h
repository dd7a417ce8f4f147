// Development switches of the kernels (kernels_align3.hip, kernels_ell.hip, dens.h, api.hip) in one place, with a guard: every
// one of them either gives WRONG results (the NVK_ABL ablations price one part of a step by leaving it out) or
// changes what the parity tests have checked, so none may be set in the product build.  tools/build_variant.sh
// — which builds experiments beside the product into variants/ — defines NVK_VARIANT_BUILD; anything else that
// sets a switch fails to compile.  The switches themselves: tools/README.md.
#pragma once

#ifndef NVK_ABL
#define NVK_ABL 0          // ablation experiments (timing only, results are wrong)
#endif
#ifndef NVK_NO_TIEFLAG
#define NVK_NO_TIEFLAG 0   // what the tie flags cost
#endif
#ifndef NVK_PAIR_DEBUG
#define NVK_PAIR_DEBUG 0   // 1: paired arithmetic, but every lane evaluates its own density at every step; 2: + prints
#endif
#ifndef NVK_NO_PAIR
#define NVK_NO_PAIR 0      // unpaired kernels also with transition rows
#endif
#ifndef NVK_TWO_PHASE
#define NVK_TWO_PHASE 1    // 0: the one-launch form (both sweeps of a read in one wave)
#endif
#ifndef NVK_SLOTS_F
#define NVK_SLOTS_F 0      // kernels_align3.hip launcher: persistent workgroups of the forward / reverse launch
#endif
#ifndef NVK_SLOTS_R
#define NVK_SLOTS_R 0
#endif
#ifndef NVK_ELL_ABL
#define NVK_ELL_ABL 0      // kernels_ell.hip, timing only: 1 no hypothesis phase, 2 no sweeps
#endif

#if !defined(NVK_VARIANT_BUILD) &&                                                                            \
    (NVK_ABL != 0 || NVK_NO_TIEFLAG != 0 || NVK_PAIR_DEBUG != 0 || NVK_NO_PAIR != 0 || NVK_TWO_PHASE != 1 || \
     NVK_ELL_ABL != 0 ||  NVK_SLOTS_F != 0 || NVK_SLOTS_R != 0 || defined(NVK_FLAG_DEBUG) || defined(NVK_DEBUG_SWITCHES))
#error "a development switch is set in a product build (variant_switches.h): use tools/build_variant.sh"
#endif
