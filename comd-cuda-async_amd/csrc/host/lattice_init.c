/* lattice_init.c -- host-side atom storage and the synthetic input generator:
 * FCC lattice (initAtoms.c:81-124), Maxwell-Boltzmann momenta with centre-of-mass removal and exact
 * rescale (:130-198, :220-248), uniform random displacements (:204-216).  Runs once, on the CPU, on the
 * host mirror of the device slot arrays; CopyDataToGpu uploads the result. */
#include "comd_host.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

Atoms* initAtoms(LinkCell* boxes)
{
   Atoms* atoms = (Atoms*)calloc(1, sizeof(Atoms));
   const size_t slots = (size_t)boxes->nTotalBoxes * boxes->maxAtoms;
   HostAtoms* h = &atoms->h;
   h->nAtoms = boxes->nAtoms;
   h->gid = (int*)calloc(slots, sizeof(int));
   h->iSpecies = (int*)calloc(slots, sizeof(int));
   real_t** arrs[10] = { &h->rx, &h->ry, &h->rz, &h->px, &h->py, &h->pz, &h->fx, &h->fy, &h->fz, &h->e };
   for (int i = 0; i < 10; ++i) *arrs[i] = (real_t*)calloc(slots, sizeof(real_t));
   return atoms;
}

void destroyAtoms(Atoms* atoms)
{
   if (!atoms) return;
   HostAtoms* h = &atoms->h;
   free(h->gid); free(h->iSpecies);
   free(h->rx); free(h->ry); free(h->rz); free(h->px); free(h->py); free(h->pz); free(h->fx); free(h->fy); free(h->fz); free(h->e);
   free(atoms);
}

int putAtomInBox(LinkCell* boxes, Atoms* atoms, int gid, int iType, real_t x, real_t y, real_t z, real_t px, real_t py, real_t pz)
{
   const real_t xyz[3] = { x, y, z };
   int iBox = getBoxFromCoord(boxes, xyz);
   if (boxes->nAtoms[iBox] >= boxes->maxAtoms) {
      fprintf(stderr, "putAtomInBox: link cell %d is full (%d slots); raise --maxAtoms\n", iBox, boxes->maxAtoms);
      exit(-1);
   }
   int iOff = iBox * boxes->maxAtoms + boxes->nAtoms[iBox];
   if (iBox < boxes->nLocalBoxes) atoms->nLocal++;
   boxes->nAtoms[iBox]++;
   HostAtoms* h = &atoms->h;
   h->gid[iOff] = gid; h->iSpecies[iOff] = iType;
   h->rx[iOff] = x; h->ry[iOff] = y; h->rz[iOff] = z;
   h->px[iOff] = px; h->py[iOff] = py; h->pz[iOff] = pz;
   return iOff;
}

static const real_t fccBasis[4][3] = { {0.25, 0.25, 0.25}, {0.25, 0.75, 0.75}, {0.75, 0.25, 0.75}, {0.75, 0.75, 0.25} };

/* visits every lattice site owned by this rank; returns the number of sites */
typedef void (*SiteFn)(void* ctx, int gid, real_t x, real_t y, real_t z);
static int forEachSite(int ny, int nz, real_t lat, const real_t* localMin, const real_t* localMax, SiteFn fn, void* ctx)
{
   int begin[3], end[3], n = 0;
   for (int a = 0; a < 3; ++a) { begin[a] = (int)floor(localMin[a] / lat); end[a] = (int)ceil(localMax[a] / lat); }
   for (int ix = begin[0]; ix < end[0]; ++ix)
      for (int iy = begin[1]; iy < end[1]; ++iy)
         for (int iz = begin[2]; iz < end[2]; ++iz)
            for (int ib = 0; ib < 4; ++ib) {
               real_t rx = (ix + fccBasis[ib][0]) * lat;
               real_t ry = (iy + fccBasis[ib][1]) * lat;
               real_t rz = (iz + fccBasis[ib][2]) * lat;
               if (rx < localMin[0] || rx >= localMax[0]) continue;
               if (ry < localMin[1] || ry >= localMax[1]) continue;
               if (rz < localMin[2] || rz >= localMax[2]) continue;
               fn(ctx, ib + 4 * (iz + nz * (iy + ny * ix)), rx, ry, rz);
               ++n;
            }
   return n;
}

static void countSite(void* ctx, int gid, real_t x, real_t y, real_t z)
{
   LinkCell* boxes = (LinkCell*)ctx; (void)gid;
   const real_t r[3] = { x, y, z };
   boxes->nAtoms[getBoxFromCoord(boxes, r)]++;
}

int countFccLattice(int nx, int ny, int nz, real_t lat, const Domain* domain, LinkCell* boxes)
{
   (void)nx;
   memset(boxes->nAtoms, 0, (size_t)boxes->nTotalBoxes * sizeof(int));
   forEachSite(ny, nz, lat, domain->localMin, domain->localMax, countSite, boxes);
   int m = 0;
   for (int i = 0; i < boxes->nTotalBoxes; ++i) { if (boxes->nAtoms[i] > m) m = boxes->nAtoms[i]; boxes->nAtoms[i] = 0; }
   return m;
}

static void placeSite(void* ctx, int gid, real_t x, real_t y, real_t z)
{
   SimFlat* s = (SimFlat*)ctx;
   putAtomInBox(s->boxes, s->atoms, gid, 0, x, y, z, 0.0, 0.0, 0.0);
}

void createFccLattice(int nx, int ny, int nz, real_t lat, SimFlat* s)
{
   forEachSite(ny, nz, lat, s->domain->localMin, s->domain->localMax, placeSite, s);
   startTimer(commReduceTimer);
   addIntParallel(&s->atoms->nLocal, &s->atoms->nGlobal, 1);
   stopTimer(commReduceTimer);
   if (s->atoms->nGlobal != 4 * nx * ny * nz) {
      fprintf(stderr, "createFccLattice: %d atoms placed, expected %d\n", s->atoms->nGlobal, 4 * nx * ny * nz);
      exit(-1);
   }
}

#define FOR_LOCAL_ATOMS(s, iOff)                                                  \
   for (int iBox_ = 0; iBox_ < (s)->boxes->nLocalBoxes; ++iBox_)                   \
      for (int iOff = (s)->boxes->maxAtoms * iBox_, end_ = iOff + (s)->boxes->nAtoms[iBox_]; iOff < end_; ++iOff)

void kineticEnergyHost(SimFlat* s)
{
   real_t eLocal[2] = { s->ePotential, 0.0 }, eSum[2];
   HostAtoms* h = &s->atoms->h;
   FOR_LOCAL_ATOMS(s, iOff) {
      real_t invMass = 0.5 / s->species[h->iSpecies[iOff]].mass;
      eLocal[1] += (h->px[iOff]*h->px[iOff] + h->py[iOff]*h->py[iOff] + h->pz[iOff]*h->pz[iOff]) * invMass;
   }
   startTimer(commReduceTimer);
   addRealParallel(eLocal, eSum, 2);
   stopTimer(commReduceTimer);
   s->ePotential = eSum[0]; s->eKinetic = eSum[1];
}

static void computeVcm(SimFlat* s, real_t vcm[3])
{
   real_t loc[4] = { 0., 0., 0., 0. }, sum[4];
   HostAtoms* h = &s->atoms->h;
   FOR_LOCAL_ATOMS(s, iOff) {
      loc[0] += h->px[iOff]; loc[1] += h->py[iOff]; loc[2] += h->pz[iOff];
      loc[3] += s->species[h->iSpecies[iOff]].mass;
   }
   startTimer(commReduceTimer);
   addRealParallel(loc, sum, 4);
   stopTimer(commReduceTimer);
   for (int a = 0; a < 3; ++a) vcm[a] = sum[a] / sum[3];
}

static void setVcm(SimFlat* s, const real_t newVcm[3])
{
   real_t oldVcm[3];
   computeVcm(s, oldVcm);
   const real_t shift[3] = { newVcm[0] - oldVcm[0], newVcm[1] - oldVcm[1], newVcm[2] - oldVcm[2] };
   HostAtoms* h = &s->atoms->h;
   FOR_LOCAL_ATOMS(s, iOff) {
      real_t mass = s->species[h->iSpecies[iOff]].mass;
      h->px[iOff] += mass * shift[0]; h->py[iOff] += mass * shift[1]; h->pz[iOff] += mass * shift[2];
   }
}

void setTemperature(SimFlat* s, real_t temperature)
{
   HostAtoms* h = &s->atoms->h;
   FOR_LOCAL_ATOMS(s, iOff) {
      real_t mass = s->species[h->iSpecies[iOff]].mass;
      real_t sigma = sqrt(kB_eV * temperature / mass);
      uint64_t seed = mkSeed((uint32_t)h->gid[iOff], 123);
      h->px[iOff] = mass * sigma * gasdev(&seed);
      h->py[iOff] = mass * sigma * gasdev(&seed);
      h->pz[iOff] = mass * sigma * gasdev(&seed);
   }
   if (temperature == 0.0) return;
   const real_t vZero[3] = { 0., 0., 0. };
   setVcm(s, vZero);
   kineticEnergyHost(s);
   real_t temp = (s->eKinetic / s->atoms->nGlobal) / kB_eV / 1.5;
   real_t scaleFactor = sqrt(temperature / temp);
   FOR_LOCAL_ATOMS(s, iOff) { h->px[iOff] *= scaleFactor; h->py[iOff] *= scaleFactor; h->pz[iOff] *= scaleFactor; }
   kineticEnergyHost(s);
}

void randomDisplacements(SimFlat* s, real_t delta)
{
   HostAtoms* h = &s->atoms->h;
   FOR_LOCAL_ATOMS(s, iOff) {
      uint64_t seed = mkSeed((uint32_t)h->gid[iOff], 457);
      h->rx[iOff] += (2.0 * lcg61(&seed) - 1.0) * delta;
      h->ry[iOff] += (2.0 * lcg61(&seed) - 1.0) * delta;
      h->rz[iOff] += (2.0 * lcg61(&seed) - 1.0) * delta;
   }
}
