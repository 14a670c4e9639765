// Stand-in for the reference project's "reproject.hpp" when integration/reproject_hip.cpp
// is compiled outside the reference tree: the declarations-only form of
// include/lens_reproject.hpp (same types and prototypes as reference src/reproject.hpp:7-27).
#pragma once
#define LRP_DECLARATIONS_ONLY
#include "lens_reproject.hpp"
