#pragma once
// Timing-only experiment switches compile kernels that KNOWINGLY return wrong results (what a stage costs without one of
// its parts).  A stray -D must never produce a library that answers wrongly without saying so: such a build needs
// -DTDM_EXPERIMENT beside the switch, and tdm_version() of an experiment build is negative (every loader checks it).
#if (defined(TDM_FINISH_NOATAN) || defined(TDM_LP2_ONEPHASE) || defined(TDM_LP2_MEMONLY) || defined(TDM_LP2_FAKE_LOADS) || \
     defined(TDM_TETRA_NOSPLIT) || defined(TDM_PFB_NOSTORE)) && !defined(TDM_EXPERIMENT)
#error "timing-only switch (wrong results) without -DTDM_EXPERIMENT"
#endif
