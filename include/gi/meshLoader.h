// include/gi/meshLoader.h -- declarations of the reference's mesh loader (include/meshLoader.h).  The reference's own meshLoader.cpp compiles
// against these headers unchanged (tests/test_reference_callers.py); include "builtin_loaders.h" instead to get the loader of this package.
#pragma once
#include <vector>
#include "octree.h"
void loadOBJ(Octree* o, const char* fname, gi::dvec3 pos, gi::dvec3 rotation, const Material& material);
void loadOBJ(Octree* o, const char* fname, gi::dvec3 pos, gi::dvec3 rotation, std::vector<const Material*> materials);
