// include/gi/photonMap.h -- drop-in for PhotonMap (include/photonMap.h:13-49): reserve / push_back / rebuild keep their
// meaning; getInRange (the candidate query of samplePhotons) is the gather kernel on the GPU.
#pragma once
#include <vector>
#include "octree.h"
#include "photon.h"

class PhotonMap {
  public:
    PhotonMap(gi::dvec3 = gi::dvec3(0, 0, 0), gi::dvec3 = gi::dvec3(0, 0, 0)) {}
    void reserve(int n) { _flat.reserve((size_t)n * 9); }
    void push_back(Photon* p)
    {
        const double v[9] = {p->origin.x, p->origin.y, p->origin.z, p->dir.x, p->dir.y, p->dir.z, p->col.x, p->col.y, p->col.z};
        _flat.insert(_flat.end(), v, v + 9);
    }
    void push_back_flat(const double* photons, int n) { _flat.insert(_flat.end(), photons, photons + (size_t)n * 9); }
    int size() const { return (int)(_flat.size() / 9); }
    // PhotonMap::rebuild (include/photonMap.cpp:33-47) inside the scene's root box (RayTracer::setScene, include/raytracer.h:38)
    void rebuild(Octree* scene)
    {
        if (gih_build_photon_map(scene->handle(), size(), _flat.data()) != 0) throw std::runtime_error(gih_last_error(scene->handle()));
        valid = true;
    }
    bool valid = false;

  private:
    std::vector<double> _flat;
};
