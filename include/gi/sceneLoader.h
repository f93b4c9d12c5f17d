// include/gi/sceneLoader.h -- declaration of the reference's scene loader (include/sceneLoader.h).  The reference's own sceneLoader.cpp compiles
// against these headers unchanged (tests/test_reference_callers.py); include "builtin_loaders.h" instead to get the loader of this package.
#pragma once
#include "octree.h"
#include "raytracer.h"
void loadScene(Octree* o, RayTracer& r, const char* fname);
