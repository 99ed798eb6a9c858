#pragma once
// Kernel-variant switches (include/cpt_hip_debug.h, cpt_set_tuning).  In the PRODUCT build (libcpt_hip.so) every switch is a compile-time
// constant holding the shipped configuration: the library carries no process-global state that changes which kernel runs, the compiler drops
// the code behind the losing settings, and cpt_set_tuning refuses every key.  The development build (-DCPT_ABLATION, libcpt_hip_abl.so;
// CPT_AMD_ABLATION=1 selects it in cpt_amd._lib) keeps them as process-global ints for same-box A/B measurements and the bit-identity tests
// between variants.
#ifdef CPT_ABLATION
#define CPT_SWITCH(decl, dflt) decl = dflt
#define CPT_SWITCH_SET(stmt) do { stmt; } while (0)
#else
#define CPT_SWITCH(decl, dflt) constexpr decl = dflt
#define CPT_SWITCH_SET(stmt) do { } while (0)
#endif

