// shim: Boost.Preprocessor is not installed.  build_manifold.hpp only uses it INSIDE the MTK_BUILD_MANIFOLD macro, which
// oracle/ref_ikfom.cpp replaces by the hand-written expansion of the three manifolds use-ikfom.hpp declares (assembled
// from the reference's own per-entry macros MTK_BOXPLUS / MTK_OPLUS / ...).
#pragma once
