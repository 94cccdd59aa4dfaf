// comd_device.hip -- launch wrappers and device-memory management behind include/comd_hip.h.
//
// Stands where the reference's gpu_kernels.cu (extern "C" wrappers, :69-660, :1013-1059) and the device half of
// gpu_utility.c (:32-347, :432-653) do.  gfx950 only; built with hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <map>
#include <utility>

#include "comd_hip.h"
#include "device_common.h"
#include "lj_kernels.h"
#include "eam_kernels.h"
#include "step_kernels.h"
#include "nl_kernels.h"
#include "eam_brick_kernels.h"
#include "eam_atom_brick_kernels.h"

static int g_rank = 0;

#define HIP_CHECK(cmd)                                                                                     \
   do {                                                                                                    \
      hipError_t status_ = (cmd);                                                                          \
      if (status_ != hipSuccess) {                                                                         \
         int dev_ = -1; (void)hipGetDevice(&dev_);                                                         \
         fprintf(stderr, "Rank %d, GPU: %d, Error in file %s at line %d\n", g_rank, dev_, __FILE__, __LINE__); \
         fprintf(stderr, "HIP error %d: %s\n", (int)status_, hipGetErrorString(status_));                  \
         exit(-1);                                                                                         \
      }                                                                                                    \
   } while (0)

#define LAUNCH_CHECK() HIP_CHECK(hipGetLastError())

static inline hipStream_t S(comdStream_t s) { return (hipStream_t)s; }
static inline int ceilDiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (device, kernel): remember the largest size asked for per pair
static void allowDynamicLds(const void* fn, size_t lds)
{
   static std::map<std::pair<int, const void*>, size_t> granted;
   int dev = 0; HIP_CHECK(hipGetDevice(&dev));
   size_t& g = granted[std::make_pair(dev, fn)];
   if (lds > g) { HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); g = lds; }
}


// ---- force-kernel timing (bench.py roofline leg): per simulation, SimGpu.timing -----------------------------------
// Two classes of launches are timed apart: kind 0 the force kernels proper (the kernel the roofline object names), kind 1 what a force
// evaluation launches beside them (LJ thread_atom: LJ_PackPositions + LJ_WaveCandidates; the cell marks of a list launch).  A force
// EVALUATION costs the sum of the two.
struct ForceTiming {
   struct Ev { hipEvent_t a, b; int kind; };
   std::vector<Ev> pool;
   size_t used = 0;
   double ms[2] = { 0.0, 0.0 };
   int launches[2] = { 0, 0 };
   bool on = false;
};

static void timingFlush(ForceTiming* t)
{
   if (!t) return;
   for (size_t i = 0; i < t->used; ++i) {
      float ms = 0.f;
      HIP_CHECK(hipEventSynchronize(t->pool[i].b));
      HIP_CHECK(hipEventElapsedTime(&ms, t->pool[i].a, t->pool[i].b));
      t->ms[t->pool[i].kind] += ms;
      t->launches[t->pool[i].kind] += 1;
   }
   t->used = 0;
}

struct ForceTimer {
   ForceTiming* t; hipStream_t st; int idx;
   ForceTimer(SimGpu* sim, hipStream_t s, int kind = 0) : t((ForceTiming*)sim->timing), st(s), idx(-1)
   {
      if (!t || !t->on) return;
      if (t->used == t->pool.size()) {
         if (t->pool.size() >= 1024) timingFlush(t);
         else { hipEvent_t a, b; HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b)); t->pool.push_back({a, b, 0}); }
      }
      idx = (int)t->used++;
      t->pool[idx].kind = kind;
      HIP_CHECK(hipEventRecord(t->pool[idx].a, st));
   }
   ~ForceTimer() { if (idx >= 0) HIP_CHECK(hipEventRecord(t->pool[idx].b, st)); }
};

extern "C" void comdForceTimingEnable(SimGpu* sim, int on)
{
   if (!sim->timing) { if (!on) return; sim->timing = new ForceTiming(); }
   ((ForceTiming*)sim->timing)->on = on != 0;
}
extern "C" void comdForceTimingReset(SimGpu* sim)
{
   ForceTiming* t = (ForceTiming*)sim->timing;
   if (!t) return;
   timingFlush(t); t->ms[0] = t->ms[1] = 0.0; t->launches[0] = t->launches[1] = 0;
}
extern "C" double comdForceTimingTotalMs(SimGpu* sim, int* nLaunches)
{
   ForceTiming* t = (ForceTiming*)sim->timing;
   if (!t) { if (nLaunches) *nLaunches = 0; return 0.0; }
   timingFlush(t);
   if (nLaunches) *nLaunches = t->launches[0];
   return t->ms[0];
}
extern "C" double comdForceTimingAuxMs(SimGpu* sim, int* nLaunches)
{
   ForceTiming* t = (ForceTiming*)sim->timing;
   if (!t) { if (nLaunches) *nLaunches = 0; return 0.0; }
   timingFlush(t);
   if (nLaunches) *nLaunches = t->launches[1];
   return t->ms[1];
}

extern "C" void comdDeviceMemInfo(long* freeBytes, long* totalBytes)
{
   size_t f = 0, tot = 0;
   HIP_CHECK(hipMemGetInfo(&f, &tot));
   if (freeBytes) *freeBytes = (long)f;
   if (totalBytes) *totalBytes = (long)tot;
}

extern "C" void* comdEventCreate(void) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); return (void*)e; }
extern "C" void comdEventRecord(void* ev, comdStream_t stream) { HIP_CHECK(hipEventRecord((hipEvent_t)ev, S(stream))); }
extern "C" float comdEventElapsedMs(void* start, void* stop)
{
   float ms = 0.f;
   HIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
   HIP_CHECK(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop));
   return ms;
}
extern "C" void comdEventDestroy(void* ev) { HIP_CHECK(hipEventDestroy((hipEvent_t)ev)); }
extern "C" void comdEventSynchronize(void* ev) { HIP_CHECK(hipEventSynchronize((hipEvent_t)ev)); }

// ---- device management ---------------------------------------------------------------------------------------
extern "C" int comdDeviceCount(void)
{
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) return 0;
   return n;
}

extern "C" int SetupGpu(int deviceId, int rank, int verbose)
{
   g_rank = rank;
   HIP_CHECK(hipSetDevice(deviceId));
   hipDeviceProp_t props;
   HIP_CHECK(hipGetDeviceProperties(&props, deviceId));
   if (verbose)
      printf("Rank %d: device %d = %s (%s), %d CUs, %.1f GiB\n", rank, deviceId, props.name, props.gcnArchName,
             props.multiProcessorCount, props.totalGlobalMem / 1073741824.0);
   return props.multiProcessorCount;
}

extern "C" void comdDeviceSynchronize(void) { HIP_CHECK(hipDeviceSynchronize()); }
extern "C" void comdStreamSynchronize(comdStream_t stream) { HIP_CHECK(hipStreamSynchronize(S(stream))); }
extern "C" void* comdDeviceMalloc(long bytes) { void* p = nullptr; HIP_CHECK(hipMalloc(&p, (size_t)(bytes > 0 ? bytes : 8))); return p; }
extern "C" void comdDeviceFree(void* p) { if (p) HIP_CHECK(hipFree(p)); }
extern "C" void* comdHostMallocPinned(long bytes) { void* p = nullptr; HIP_CHECK(hipHostMalloc(&p, (size_t)(bytes > 0 ? bytes : 8), hipHostMallocDefault)); return p; }
extern "C" void comdHostFreePinned(void* p) { if (p) HIP_CHECK(hipHostFree(p)); }
extern "C" void comdMemcpyHtoD(void* dst, const void* src, long bytes) { HIP_CHECK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice)); }
extern "C" void comdMemcpyDtoH(void* dst, const void* src, long bytes) { HIP_CHECK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost)); }
extern "C" void comdMemcpyDtoDAsync(void* dst, const void* src, long bytes, comdStream_t stream)
{
   HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, S(stream)));
}

extern "C" void comdMemcpyAsync(void* dst, const void* src, long bytes, int kind, comdStream_t stream)
{
   if (bytes <= 0) return;
   const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : kind == 3 ? hipMemcpyDeviceToDevice : hipMemcpyHostToHost;
   HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)bytes, k, S(stream)));
}
// cudaMemset of the reference's host files (timestep.c:224): issued on the legacy default stream, which waits for and is waited for by the reference's
// streams (created with flags 0, gpu_utility.c:150-152).  The -a 1 streams here are non-blocking, so the same ordering is spelled out.
extern "C" void comdDeviceMemset(void* p, int value, long bytes)
{
   if (bytes <= 0) return;
   HIP_CHECK(hipDeviceSynchronize());
   HIP_CHECK(hipMemset(p, value, (size_t)bytes));
   HIP_CHECK(hipDeviceSynchronize());
}

template <typename T> static T* dalloc(size_t n, bool zero = true)
{
   T* p = nullptr;
   HIP_CHECK(hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)));
   if (zero) HIP_CHECK(hipMemset(p, 0, (n ? n : 1) * sizeof(T)));
   return p;
}

static void uploadTable(InterpolationObjectGpu* t, int n, real_t x0, real_t invDx, const real_t* hostValues)
{
   t->n = n; t->x0 = x0; t->invDx = invDx;
   t->xn = x0 + n / invDx;                     // gpu_utility.c:446, 460-461
   t->invDxHalf = invDx * 0.5;
   t->invDxXx0 = x0 * invDx;
   t->values = dalloc<real_t>((size_t)n + 3, false);
   HIP_CHECK(hipMemcpy(t->values, hostValues, ((size_t)n + 3) * sizeof(real_t), hipMemcpyHostToDevice));
}

extern "C" void initLinkCellsGpu(LinkCellGpu* b, const GpuConfig* cfg)
{
   b->nLocalBoxes = cfg->nLocalBoxes; b->nTotalBoxes = cfg->nTotalBoxes;
   for (int a = 0; a < 3; ++a) {
      b->gridSize[a] = cfg->gridSize[a]; b->localMin[a] = cfg->localMin[a]; b->localMax[a] = cfg->localMax[a];
      b->invBoxSize[a] = 1.0 / cfg->boxSize[a];
   }
   b->nAtoms = dalloc<int>(cfg->nTotalBoxes);
   b->boxIDLookUp = b->boxIDLookUpReverse = nullptr;
   if (cfg->boxIDLookUp && cfg->boxIDLookUpReverse) {     // gpu_utility.c:594-595
      b->boxIDLookUp = dalloc<int>(cfg->nLocalBoxes, false);
      b->boxIDLookUpReverse = dalloc<int>(cfg->nLocalBoxes, false);
      HIP_CHECK(hipMemcpy(b->boxIDLookUp, cfg->boxIDLookUp, (size_t)cfg->nLocalBoxes * sizeof(int), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(b->boxIDLookUpReverse, cfg->boxIDLookUpReverse, (size_t)cfg->nLocalBoxes * sizeof(int), hipMemcpyHostToDevice));
   }
}

extern "C" void AllocateGpu(SimGpu* sim, const GpuConfig* cfg)
{
   memset(sim, 0, sizeof(*sim));
   HIP_CHECK(hipGetDevice(&sim->deviceId));
   sim->rank = cfg->rank; g_rank = cfg->rank;
   sim->maxAtoms = cfg->maxAtoms;
   sim->needEnergy = 1;
   sim->latticeConstant = cfg->latticeConstant;
   sim->do_eam = cfg->do_eam;
   sim->mass = cfg->mass;
   if (cfg->maxAtoms < 1 || cfg->maxAtoms > 1024) { fprintf(stderr, "AllocateGpu: maxAtoms %d outside [1,1024]\n", cfg->maxAtoms); exit(-1); }
   if (!cfg->do_eam && (cfg->maxAtoms % 64) != 0) { fprintf(stderr, "AllocateGpu: LJ needs maxAtoms %% 64 == 0 (got %d)\n", cfg->maxAtoms); exit(-1); }

   initLinkCellsGpu(&sim->boxes, cfg);
   const size_t slots = (size_t)cfg->nTotalBoxes * cfg->maxAtoms;      // 64-bit: 256^3 LJ needs 136 M slots
   AtomsGpu* at = &sim->atoms;
   at->r.x = dalloc<real_t>(slots); at->r.y = dalloc<real_t>(slots); at->r.z = dalloc<real_t>(slots);
   at->p.x = dalloc<real_t>(slots); at->p.y = dalloc<real_t>(slots); at->p.z = dalloc<real_t>(slots);
   at->f.x = dalloc<real_t>(slots); at->f.y = dalloc<real_t>(slots); at->f.z = dalloc<real_t>(slots);
   at->e = dalloc<real_t>(slots);
   at->iSpecies = dalloc<int>(slots); at->gid = dalloc<int>(slots);
   sim->neighbor_cells = dalloc<int>((size_t)cfg->nLocalBoxes * 27, false);
   HIP_CHECK(hipMemcpy(sim->neighbor_cells, cfg->neighborCells, (size_t)cfg->nLocalBoxes * 27 * sizeof(int), hipMemcpyHostToDevice));
   sim->species_mass = dalloc<real_t>(1, false);
   HIP_CHECK(hipMemcpy(sim->species_mass, &cfg->mass, sizeof(real_t), hipMemcpyHostToDevice));

   sim->lj_pot.cutoff = cfg->ljCutoff; sim->lj_pot.sigma = cfg->ljSigma; sim->lj_pot.epsilon = cfg->ljEpsilon;
   if (cfg->do_eam) {
      sim->eam_pot.cutoff = cfg->eamCutoff;
      uploadTable(&sim->eam_pot.phi, cfg->nPhi, cfg->phiX0, cfg->phiInvDx, cfg->phiValues);
      uploadTable(&sim->eam_pot.rho, cfg->nRho, cfg->rhoX0, cfg->rhoInvDx, cfg->rhoValues);
      uploadTable(&sim->eam_pot.f,   cfg->nF,   cfg->fX0,   cfg->fInvDx,   cfg->fValues);
      if (cfg->phiSpline && cfg->rhoSpline) {             // gpu_utility.c:247-249, 478-500
         auto up = [](InterpolationSplineObjectGpu* t, int n, real_t x0, real_t invDx, const real_t* host) {
            t->n = n; t->x0 = (float)x0; t->xn = (float)(x0 + n / invDx); t->invDx = (float)invDx; t->invDxXx0 = (float)(invDx * x0);
            t->coefficients = dalloc<real_t>((size_t)4 * n, false);
            HIP_CHECK(hipMemcpy(t->coefficients, host, (size_t)4 * n * sizeof(real_t), hipMemcpyHostToDevice));
         };
         up(&sim->eam_pot.phiS, cfg->nPhi, cfg->phiX0, cfg->phiInvDx, cfg->phiSpline);
         up(&sim->eam_pot.rhoS, cfg->nRho, cfg->rhoX0, cfg->rhoInvDx, cfg->rhoSpline);
      }
      sim->eam_pot.rhobar = dalloc<real_t>(slots);
      sim->eam_pot.dfEmbed = dalloc<real_t>(slots);
   }
   if (cfg->skinDistance > 0.0) {               // gpu_neighborList.c:49-86 initNeighborListGpu
      NeighborListGpu* nl = &at->neighborList;
      const real_t cutoff = cfg->do_eam ? cfg->eamCutoff : cfg->ljCutoff;
      nl->skinDistance = cfg->skinDistance; nl->skinDistance2 = cfg->skinDistance * cfg->skinDistance;
      nl->skinDistanceHalf2 = 0.25 * nl->skinDistance2;
      nl->maxNeighbors = cfg->maxNeighbors;
      if (nl->maxNeighbors <= 0) {
         // atoms inside the list sphere at the perfect-lattice density (4 per lat^3), + 20 % and 24 for thermal crowding
         const double rl = cutoff + cfg->skinDistance, lat = cfg->latticeConstant > 0.0 ? cfg->latticeConstant : 3.615;
         const double expect = 4.0 / (lat * lat * lat) * 4.18879020478639 * rl * rl * rl;
         nl->maxNeighbors = ((int)(1.2 * expect) + 24 + 7) / 8 * 8;
      }
      if (nl->maxNeighbors > 27 * cfg->maxAtoms) nl->maxNeighbors = 27 * cfg->maxAtoms;
      const size_t localSlots = (size_t)cfg->nLocalBoxes * cfg->maxAtoms;
      nl->slabFormat = !cfg->usePairlist && !cfg->do_eam && cfg->maxAtoms % 64 == 0 && cfg->maxAtoms <= 512 && !getenv("COMD_NL_GLOBAL");
      if (cfg->usePairlist) {
         if (cfg->do_eam) { fprintf(stderr, "AllocateGpu: pairlists (-L) are an LJ cta_cell feature\n"); exit(-1); }
         nl->slabFormat = 3;
         const int threads = cfg->maxAtoms < 256 ? cfg->maxAtoms : 256;
         nl->pairlistWaves = (threads + 63) / 64;
         nl->pairlist = dalloc<unsigned>((size_t)cfg->nLocalBoxes * nl->pairlistWaves * LJ_CTA_SLABS * LJ_PL_WORDS);
         nl->pairlistBuildId = -1;
      } else if (cfg->do_eam && !getenv("COMD_NL_GLOBAL") && !(getenv("COMD_EAM_NL") && !strcmp(getenv("COMD_EAM_NL"), "lds"))
                 && (double)cfg->nTotalBoxes * cfg->maxAtoms * sizeof(real_t) < 4294967296.0) {
         // [round 4] EAM: rows of the brick kernel (eam_brick_kernels.h, LISTED): 16-bit record numbers in the LDS image of the atom's brick, kept from one
         // list build to the next.  Any table size (setfl tables and -P coefficients are read through L2), any cell capacity.
         nl->slabFormat = 4;
         nl->brickRowLen = (nl->maxNeighbors + 7) / 8 * 8;
         if (nl->brickRowLen > 2 * EAM_LIST_WORDS * 16) nl->brickRowLen = 2 * EAM_LIST_WORDS * 16;      // 16 lanes x 12 words x 2 numbers
         const int lanesMin = (nl->brickRowLen + 2 * EAM_LIST_WORDS - 1) / (2 * EAM_LIST_WORDS);          // as the kernel derives them from `rows`
         nl->brickRoundAtoms = 64 / lanesMin < 16 ? 64 / lanesMin : 16;
         nl->brickRounds = (cfg->maxAtoms + nl->brickRoundAtoms - 1) / nl->brickRoundAtoms;
         const int lanesPerAtom = 64 / nl->brickRoundAtoms;                                                // the fewest lanes an atom is ever dealt to
         int words = (nl->brickRowLen / 2 + lanesPerAtom - 1) / lanesPerAtom;
         if (words > EAM_LIST_WORDS) words = EAM_LIST_WORDS;
         nl->brickQuads = (words + 3) / 4;
         nl->brickRows = dalloc<unsigned>((size_t)cfg->nLocalBoxes * nl->brickRounds * nl->brickQuads * 64 * 4, false);
         nl->brickRowCount = dalloc<unsigned short>(localSlots);
      } else if (cfg->do_eam && cfg->maxAtoms <= 64 && eamCtaTableBytes(1, cfg->nRho, cfg->nPhi) <= 32 * 1024 && !getenv("COMD_NL_GLOBAL")) {
         // EAM with LDS-sized tables: 16-bit entries into the wave's staging of the whole 27-cell stencil
         nl->slabFormat = 2;
         nl->slabRows = nl->maxNeighbors;
         nl->list16 = dalloc<unsigned short>(localSlots * nl->slabRows, false);
         nl->nNeighbors = dalloc<int>(localSlots);
         nl->stats = dalloc<int>(2);
      } else if (nl->slabFormat) {
         // share of the list sphere (radius R) one group can hold.  3 groups: the atom's own x-plane of cells, thickness wx, cuts at
         // most wx * pi R^2 out of 4/3 pi R^3 (78 % for cells of about R); 9 groups: its own z-column, wx * wy * 2R (51 %)
         const double R = cutoff + cfg->skinDistance;
         double share = NL_DIAGONAL ? 0.62                     // own cell + edge cells; measured 0.44-0.5 of the list for cells of about R
                      : NL_GROUPS == 3 ? 3.0 * cfg->boxSize[0] / (4.0 * R)
                                       : 3.0 * cfg->boxSize[0] * cfg->boxSize[1] / (2.0 * 3.14159265358979 * R * R);
         if (share > 1.0) share = 1.0;
         nl->slabRows = ((int)(share * nl->maxNeighbors) + 7) / 8 * 8;
         if (nl->slabRows > NL_GROUP_CELLS * cfg->maxAtoms) nl->slabRows = NL_GROUP_CELLS * cfg->maxAtoms;
         nl->list16 = dalloc<unsigned short>(localSlots * NL_GROUPS * nl->slabRows, false);
         nl->nNeighbors = dalloc<int>(localSlots * NL_GROUPS);
         nl->stats = dalloc<int>(2);
      } else {
         nl->list = dalloc<int>(localSlots * nl->maxNeighbors, false);
         nl->nNeighbors = dalloc<int>(localSlots);
      }
      nl->lastR.x = dalloc<real_t>(localSlots); nl->lastR.y = dalloc<real_t>(localSlots); nl->lastR.z = dalloc<real_t>(localSlots);
      HIP_CHECK(hipHostMalloc((void**)&nl->updateRequiredHost, 64, hipHostMallocDefault));      // pinned: written by kernels, read by the host
      memset(nl->updateRequiredHost, 0, 64);
      HIP_CHECK(hipHostGetDevicePointer((void**)&nl->updateRequired, nl->updateRequiredHost, 0));
      nl->forceRebuildFlag = 1; nl->nBuilds = 0;
   }
   sim->nAtomsPrev = dalloc<int>(cfg->nTotalBoxes);
   sim->cellDirty = dalloc<int>(cfg->nTotalBoxes);
   sim->cellArrivals = dalloc<int>((size_t)3 * cfg->nTotalBoxes);
   sim->d_updateLinkCellsRequired = dalloc<int>(1);
   sim->status = dalloc<int>(4);
   sim->reduceBlocks = 1024;
   sim->reduceBuf = dalloc<real_t>(2 * (size_t)sim->reduceBlocks + 2);
   HIP_CHECK(hipHostMalloc((void**)&sim->pinned, 64 * sizeof(real_t), hipHostMallocDefault));
   memset(sim->pinned, 0, 64 * sizeof(real_t));
   HIP_CHECK(hipHostGetDevicePointer((void**)&sim->statusMirrorDev, (int*)(sim->pinned + 32), 0));

   if (cfg->gpuAsync) {                         // gpu_utility.c:150-159
      hipStream_t bs, is;
      int lo = 0, hi = 0;
      HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
      HIP_CHECK(hipStreamCreateWithPriority(&bs, hipStreamNonBlocking, hi));
      HIP_CHECK(hipStreamCreateWithFlags(&is, hipStreamNonBlocking));
      sim->boundary_stream = (comdStream_t)bs; sim->interior_stream = (comdStream_t)is;
   } else {
      sim->boundary_stream = nullptr; sim->interior_stream = nullptr;
   }
   HIP_CHECK(hipDeviceSynchronize());            // the zeroing above is done before any non-blocking stream can touch these arrays
}

extern "C" void SetBoundaryCells(SimGpu* sim, int nBoundary, const int* boundary, int nInterior, const int* interior,
                                 int nBoundary1, const int* boundary1)
{
   sim->n_boundary_cells = nBoundary; sim->n_interior_cells = nInterior; sim->n_boundary1_cells = nBoundary1;
   sim->boundary_cells = dalloc<int>(nBoundary, false);
   sim->interior_cells = dalloc<int>(nInterior, false);
   sim->boundary1_cells = dalloc<int>(nBoundary1, false);
   if (nBoundary)  HIP_CHECK(hipMemcpy(sim->boundary_cells, boundary, (size_t)nBoundary * sizeof(int), hipMemcpyHostToDevice));
   if (nInterior)  HIP_CHECK(hipMemcpy(sim->interior_cells, interior, (size_t)nInterior * sizeof(int), hipMemcpyHostToDevice));
   if (nBoundary1) HIP_CHECK(hipMemcpy(sim->boundary1_cells, boundary1, (size_t)nBoundary1 * sizeof(int), hipMemcpyHostToDevice));
}

extern "C" void CopyDataToGpu(SimGpu* sim, const HostAtoms* h)
{
   const size_t slots = (size_t)sim->boxes.nTotalBoxes * sim->maxAtoms;
   sim->max_atoms_cell = 0;                       // unknown until the next updateNAtomsCpu: launch cap/64 waves per cell
   HIP_CHECK(hipMemcpy(sim->boxes.nAtoms, h->nAtoms, (size_t)sim->boxes.nTotalBoxes * sizeof(int), hipMemcpyHostToDevice));
   HIP_CHECK(hipMemcpy(sim->atoms.gid, h->gid, slots * sizeof(int), hipMemcpyHostToDevice));
   HIP_CHECK(hipMemcpy(sim->atoms.iSpecies, h->iSpecies, slots * sizeof(int), hipMemcpyHostToDevice));
   HIP_CHECK(hipMemcpy(sim->atoms.r.x, h->rx, slots * sizeof(real_t), hipMemcpyHostToDevice));
   HIP_CHECK(hipMemcpy(sim->atoms.r.y, h->ry, slots * sizeof(real_t), hipMemcpyHostToDevice));
   HIP_CHECK(hipMemcpy(sim->atoms.r.z, h->rz, slots * sizeof(real_t), hipMemcpyHostToDevice));
   HIP_CHECK(hipMemcpy(sim->atoms.p.x, h->px, slots * sizeof(real_t), hipMemcpyHostToDevice));
   HIP_CHECK(hipMemcpy(sim->atoms.p.y, h->py, slots * sizeof(real_t), hipMemcpyHostToDevice));
   HIP_CHECK(hipMemcpy(sim->atoms.p.z, h->pz, slots * sizeof(real_t), hipMemcpyHostToDevice));
}

extern "C" void GetDataFromGpu(SimGpu* sim, HostAtoms* h)
{
   HIP_CHECK(hipDeviceSynchronize());
   const size_t slots = (size_t)sim->boxes.nTotalBoxes * sim->maxAtoms;
   HIP_CHECK(hipMemcpy(h->nAtoms, sim->boxes.nAtoms, (size_t)sim->boxes.nTotalBoxes * sizeof(int), hipMemcpyDeviceToHost));
   HIP_CHECK(hipMemcpy(h->gid, sim->atoms.gid, slots * sizeof(int), hipMemcpyDeviceToHost));
   HIP_CHECK(hipMemcpy(h->iSpecies, sim->atoms.iSpecies, slots * sizeof(int), hipMemcpyDeviceToHost));
   real_t* dst[10] = { h->rx, h->ry, h->rz, h->px, h->py, h->pz, h->fx, h->fy, h->fz, h->e };
   real_t* src[10] = { sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z, sim->atoms.p.x, sim->atoms.p.y, sim->atoms.p.z,
                       sim->atoms.f.x, sim->atoms.f.y, sim->atoms.f.z, sim->atoms.e };
   for (int i = 0; i < 10; ++i) if (dst[i]) HIP_CHECK(hipMemcpy(dst[i], src[i], slots * sizeof(real_t), hipMemcpyDeviceToHost));
}

extern "C" void updateNAtomsCpu(SimGpu* sim, int* nAtomsHost)
{
   HIP_CHECK(hipDeviceSynchronize());
   HIP_CHECK(hipMemcpy(nAtomsHost, sim->boxes.nAtoms, (size_t)sim->boxes.nTotalBoxes * sizeof(int), hipMemcpyDeviceToHost));
   int m = 0;
   for (int i = 0; i < sim->boxes.nTotalBoxes; ++i) if (nAtomsHost[i] > m) m = nAtomsHost[i];
   sim->max_atoms_cell = m;                       // gpu_types.h:160 max_atoms_cell: sizes the LJ thread_atom workgroups
}

// gpu_utility.h:60-69, the rest of the staging surface (the reference's cpu_nl path moves atoms between its host arrays and the device with these)
extern "C" void cudaCopyDtH(void* dst, const void* src, int size) { HIP_CHECK(hipMemcpy(dst, src, (size_t)size, hipMemcpyDeviceToHost)); }      // gpu_utility.c:46-49, same signature

extern "C" void GetLocalAtomsFromGpu(SimGpu* sim, HostAtoms* h)                     // gpu_utility.c:656-673: momenta, positions and gids of the LOCAL cells
{
   HIP_CHECK(hipDeviceSynchronize());
   const size_t slots = (size_t)sim->boxes.nLocalBoxes * sim->maxAtoms;
   real_t* dst[6] = { h->px, h->py, h->pz, h->rx, h->ry, h->rz };
   real_t* src[6] = { sim->atoms.p.x, sim->atoms.p.y, sim->atoms.p.z, sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z };
   for (int i = 0; i < 6; ++i) if (dst[i]) HIP_CHECK(hipMemcpy(dst[i], src[i], slots * sizeof(real_t), hipMemcpyDeviceToHost));
   if (h->gid) HIP_CHECK(hipMemcpy(h->gid, sim->atoms.gid, slots * sizeof(int), hipMemcpyDeviceToHost));
}

extern "C" void updateGpuHalo(SimGpu* sim, const HostAtoms* h)                       // gpu_utility.c:714-757: the HALO cells' slots, host -> device
{
   const size_t first = (size_t)sim->boxes.nLocalBoxes * sim->maxAtoms, slots = (size_t)(sim->boxes.nTotalBoxes - sim->boxes.nLocalBoxes) * sim->maxAtoms;
   const real_t* src[6] = { h->px, h->py, h->pz, h->rx, h->ry, h->rz };
   real_t* dst[6] = { sim->atoms.p.x, sim->atoms.p.y, sim->atoms.p.z, sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z };
   for (int i = 0; i < 6; ++i) HIP_CHECK(hipMemcpy(dst[i] + first, src[i] + first, slots * sizeof(real_t), hipMemcpyHostToDevice));
   HIP_CHECK(hipMemcpy(sim->atoms.gid + first, h->gid + first, slots * sizeof(int), hipMemcpyHostToDevice));
   HIP_CHECK(hipMemcpy(sim->atoms.iSpecies + first, h->iSpecies + first, slots * sizeof(int), hipMemcpyHostToDevice));
}

extern "C" void updateNAtomsGpu(SimGpu* sim, const int* nAtomsHost)                  // gpu_utility.c:602-605
{
   HIP_CHECK(hipMemcpy(sim->boxes.nAtoms, nAtomsHost, (size_t)sim->boxes.nTotalBoxes * sizeof(int), hipMemcpyHostToDevice));
}

// gpu_utility.c:678-712, a host loop there as here: the atoms of the halo cells, cell by cell, as one SoA message in h_compactAtoms (no header);
// h_cellOffset[i] = atoms in the halo cells before the i-th, nHalo + 1 entries.  (The reference zeroes h_cellOffset[nLocalBoxes] instead of [0] and
// relies on the caller's calloc; entry 0 is written here.)
extern "C" int compactHaloCells(const HostAtoms* h, int nLocalBoxes, int nTotalBoxes, int maxAtoms, char* h_compactAtoms, int* h_cellOffset)
{
   const int nHalo = nTotalBoxes - nLocalBoxes;
   h_cellOffset[0] = 0;
   for (int i = 0; i < nHalo; ++i) h_cellOffset[i + 1] = h_cellOffset[i] + h->nAtoms[nLocalBoxes + i];
   const int n = h_cellOffset[nHalo];
   int* gid = (int*)h_compactAtoms; int* type = gid + n;
   real_t* m = (real_t*)(type + n);
   for (int i = 0; i < nHalo; ++i) {
      size_t o = (size_t)(nLocalBoxes + i) * maxAtoms;
      for (int k = h_cellOffset[i]; k < h_cellOffset[i + 1]; ++k, ++o) {
         gid[k] = h->gid[o]; type[k] = h->iSpecies[o];
         m[k] = h->rx[o]; m[(size_t)n + k] = h->ry[o]; m[2 * (size_t)n + k] = h->rz[o];
         m[3 * (size_t)n + k] = h->px[o]; m[4 * (size_t)n + k] = h->py[o]; m[5 * (size_t)n + k] = h->pz[o];
      }
   }
   return n;
}

extern "C" void DestroyGpu(SimGpu* sim)
{
   HIP_CHECK(hipDeviceSynchronize());
   if (sim->timing) {
      ForceTiming* t = (ForceTiming*)sim->timing;
      timingFlush(t);
      for (auto& ev : t->pool) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
      delete t;
   }
   void* ptrs[] = { sim->boxes.nAtoms, sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z, sim->atoms.p.x, sim->atoms.p.y, sim->atoms.p.z,
                    sim->atoms.f.x, sim->atoms.f.y, sim->atoms.f.z, sim->atoms.e, sim->atoms.iSpecies, sim->atoms.gid,
                    sim->neighbor_cells, sim->species_mass, sim->eam_pot.phi.values, sim->eam_pot.rho.values, sim->eam_pot.f.values,
                    sim->eam_pot.rhobar, sim->eam_pot.dfEmbed, sim->nAtomsPrev, sim->cellDirty, sim->cellArrivals, sim->d_updateLinkCellsRequired, sim->status, sim->reduceBuf,
                    sim->boundary_cells, sim->interior_cells, sim->boundary1_cells,
                    sim->atoms.neighborList.list, sim->atoms.neighborList.nNeighbors, sim->atoms.neighborList.lastR.x,
                    sim->atoms.neighborList.lastR.y, sim->atoms.neighborList.lastR.z,
                    sim->atoms.neighborList.list16, sim->atoms.neighborList.stats, sim->atoms.neighborList.pairlist,
                    sim->boxes.boxIDLookUp, sim->boxes.boxIDLookUpReverse, sim->eam_pot.phiS.coefficients, sim->eam_pot.rhoS.coefficients,
                    sim->eam_pot.pairRows, sim->eam_pot.pairRowCount, sim->eam_pot.cellSel, sim->eam_pot.brickGroup, sim->eam_pot.brickList, sim->eam_pot.brickSel, sim->eam_pot.brickStats, sim->eam_pot.atomRows, sim->eam_pot.atomRowCount, sim->eam_pot.atomBrickSel,
                    sim->atoms.neighborList.brickRows, sim->atoms.neighborList.brickRowCount, sim->adapterScan, sim->lj_pot.waveCand, sim->lj_pot.waveCandCount, sim->lj_pot.packedR[0], sim->lj_pot.packedR[1], sim->lj_pot.packedF[0], sim->lj_pot.packedF[1] };
   for (void* p : ptrs) if (p) HIP_CHECK(hipFree(p));
   if (sim->statusEvent) (void)hipEventDestroy((hipEvent_t)sim->statusEvent);
   if (sim->pinned) HIP_CHECK(hipHostFree(sim->pinned));
   if (sim->atoms.neighborList.updateRequiredHost) HIP_CHECK(hipHostFree(sim->atoms.neighborList.updateRequiredHost));
   if (sim->atoms.neighborList.brickStatsMirror) HIP_CHECK(hipHostFree(sim->atoms.neighborList.brickStatsMirror));
   if (sim->atoms.neighborList.brickStatsEvent) (void)hipEventDestroy((hipEvent_t)sim->atoms.neighborList.brickStatsEvent);
   if (sim->boundary_stream) HIP_CHECK(hipStreamDestroy(S(sim->boundary_stream)));
   if (sim->interior_stream) HIP_CHECK(hipStreamDestroy(S(sim->interior_stream)));
   memset(sim, 0, sizeof(*sim));
}

extern "C" void emptyHaloCellsGpu(SimGpu* sim, comdStream_t stream)
{
   const int nHalo = sim->boxes.nTotalBoxes - sim->boxes.nLocalBoxes;
   HIP_CHECK(hipMemsetAsync(sim->boxes.nAtoms + sim->boxes.nLocalBoxes, 0, (size_t)nHalo * sizeof(int), S(stream)));
}

extern "C" void comdCheckStatus(SimGpu* sim, const char* where)
{
   int st[4];
   HIP_CHECK(hipDeviceSynchronize());
   HIP_CHECK(hipMemcpy(st, sim->status, sizeof st, hipMemcpyDeviceToHost));
   if (st[0] | st[1] | st[2] | st[3]) {
      fprintf(stderr, "Rank %d, GPU: %d, %s: ", g_rank, sim->deviceId, where);
      if (st[0] & 1) fprintf(stderr, "a link cell overflowed its %d slots (raise --maxAtoms); ", sim->maxAtoms);
      if (st[0] & 2) fprintf(stderr, "a cell stencil holds more atoms than the cta_cell kernel can stage; ");
      if (st[1])     fprintf(stderr, "an atom moved beyond the halo region and was lost; ");
      if (st[2])     fprintf(stderr, "a halo message overflowed its buffer, or grew by more than 12.5 %% + 64 atoms in one step (COMD_HALO_HANDSHAKE=1 exchanges exact sizes); ");
      if (st[3] & 1) fprintf(stderr, "an atom has more neighbours inside the cutoff than a row of the EAM cta_cell kernel holds (1.5 x the FCC count; use -m thread_atom); ");
      if (st[3] & 2) fprintf(stderr, "an atom has more than %d neighbours inside cutoff + skin (raise --maxNeighbors); ", sim->atoms.neighborList.maxNeighbors);
      if (st[3] & 4) fprintf(stderr, "eamForce3Gpu[Async] covered the cells with another partition than eamForce1Gpu[Async] (include/comd_hip.h: same lists in both passes); ");
      fprintf(stderr, "\n");
      exit(-1);
   }
}

extern "C" void comdPollStatus(SimGpu* sim, comdStream_t stream, const char* where)
{
   int* mirror = (int*)(sim->pinned + 32);               // four ints of the 64-real_t pinned block, away from the energy words
   // [round 4] The fused drift kernels copy the status words into the mirror as they start (step_kernels.h skinProgress): nothing to enqueue here -- the separate
   // 16-byte copy per step was a blit kernel with a launch gap on either side, 10 us of every step.  Without such a kernel since the last poll: the copy, as before.
   if (sim->statusMirrored) {
      sim->statusMirrored = 0;
      if (mirror[0] | mirror[1] | mirror[2] | mirror[3]) comdCheckStatus(sim, where);
      return;
   }
   if (sim->statusEvent) {
      if (hipEventQuery((hipEvent_t)sim->statusEvent) != hipSuccess) return;       // the previous mirror has not landed yet: look again next step
      if (mirror[0] | mirror[1] | mirror[2] | mirror[3]) comdCheckStatus(sim, where);
   } else {
      hipEvent_t e; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); sim->statusEvent = (void*)e;
   }
   HIP_CHECK(hipMemcpyAsync(mirror, sim->status, 4 * sizeof(int), hipMemcpyDeviceToHost, S(stream)));
   HIP_CHECK(hipEventRecord((hipEvent_t)sim->statusEvent, S(stream)));
}

extern "C" int comdReadDeviceInt(const int* d_ptr, comdStream_t stream)
{
   int v = 0;
   HIP_CHECK(hipMemcpyAsync(&v, d_ptr, sizeof(int), hipMemcpyDeviceToHost, S(stream)));
   HIP_CHECK(hipStreamSynchronize(S(stream)));
   return v;
}

// ---- force -------------------------------------------------------------------------------------------------------
static NlView nlView(SimGpu* sim)
{
   NeighborListGpu* n = &sim->atoms.neighborList;
   if (!n->list && !n->list16) { fprintf(stderr, "the *_nl methods need Verlet lists: allocate with GpuConfig.skinDistance > 0\n"); exit(-1); }
   if (n->nBuilds == 0) { fprintf(stderr, "the *_nl methods need buildNeighborListGpu before the first force call\n"); exit(-1); }
   NlView v; v.list = n->list; v.count = n->nNeighbors; v.maxNbr = n->maxNeighbors;
   return v;
}

static LjArgs makeLjArgs(SimGpu* sim, int num_cells, int* cells_list)
{
   LjArgs a;
   a.rx = sim->atoms.r.x; a.ry = sim->atoms.r.y; a.rz = sim->atoms.r.z;
   a.fx = sim->atoms.f.x; a.fy = sim->atoms.f.y; a.fz = sim->atoms.f.z; a.e = sim->atoms.e;
   a.nAtoms = sim->boxes.nAtoms; a.nbr = sim->neighbor_cells; a.cells = cells_list;
   a.nCells = num_cells; a.cap = sim->maxAtoms;
   const double sigma = sim->lj_pot.sigma, rc = sim->lj_pot.cutoff;
   a.rc2 = rc * rc;
   a.s6 = sigma * sigma * sigma * sigma * sigma * sigma;
   a.s6x2 = 2.0 * (sigma * sigma * sigma * sigma * sigma * sigma);
   const double rc6 = a.s6 / (a.rc2 * a.rc2 * a.rc2);
   a.eShift = rc6 * (rc6 - 1.0);                 // POT_SHIFT 1.0
   a.eps = sim->lj_pot.epsilon;
   return a;
}

// Per-atom energies are consumed only by computeEnergy.  The host announces with comdSetEnergyNeeded(0) that the coming
// force evaluations feed no energy read (all but the last step of a timestep() call); the default is 1 (always compute).
extern "C" void comdSetEnergyNeeded(SimGpu* sim, int on) { sim->needEnergy = on; }

// Margins of the point-to-box pruning test (LJ_WaveCandidates, LJ_Force_cta_cell_boxes): the test may keep a candidate it need not, never drop one a
// lane would accept.  A relative margin on rc^2 and on the half widths covers the arithmetic of the distance; the box CENTRE is formed from absolute
// coordinates, whose ulp grows with the box (float: 3e-5 A at 289 A, 1.2e-4 A beyond 1024 A -- ADVICE r2), so the cutoff is also pushed out by eight
// ulps of the largest coordinate this rank can see (local domain + one halo cell), per axis.
static void ljBoxMargins(const SimGpu* sim, real_t rc2, real_t* rc2Box, real_t* grow)
{
   const double rel = sizeof(real_t) == 8 ? 1e-12 : 1e-5, eps = sizeof(real_t) == 8 ? 2.220446049250313e-16 : 1.1920929e-07;
   double big = 0.0;
   for (int a = 0; a < 3; ++a) {
      const double cellW = 1.0 / sim->boxes.invBoxSize[a];
      const double lo = fabs((double)sim->boxes.localMin[a] - cellW), hi = fabs((double)sim->boxes.localMax[a] + cellW);
      if (lo > big) big = lo;
      if (hi > big) big = hi;
   }
   const double rc = sqrt((double)rc2) * (1.0 + rel) + 8.0 * 1.7320508 * eps * big;
   *rc2Box = (real_t)(rc * rc);
   *grow = (real_t)(1.0 + rel);
}

// The list build of thread_atom (LJ_WaveCandidates) makes the same test in SINGLE precision on positions relative to the corner of the local domain
// (LJ_PackPositions): the conversion moves a coordinate by half an ulp of the largest relative coordinate (local domain + one halo cell on either side),
// the arithmetic of the box distance is good to ~1e-6 relative -- the cutoff is pushed out by 1e-5 relative + eight such ulps per axis.
static void ljBoxMarginsF(const SimGpu* sim, real_t rc2, float* rc2Box, float* grow)
{
   const double rel = 1e-5, eps = 1.1920929e-07;
   double big = 0.0;
   for (int a = 0; a < 3; ++a) {
      const double ext = (double)sim->boxes.localMax[a] - (double)sim->boxes.localMin[a] + 2.0 / sim->boxes.invBoxSize[a];
      if (ext > big) big = ext;
   }
   const double rc = sqrt((double)rc2) * (1.0 + rel) + 8.0 * 1.7320508 * eps * big;
   *rc2Box = (float)(rc * rc * (1.0 + 2.0 * eps));
   *grow = (float)(1.0 + rel);
}

// thread_atom (the BASELINE-named kernel): candidate lists, then the force kernel
static void launchLjThreadAtom(SimGpu* sim, const LjArgs& a, int num_cells, int* cells_list, comdStream_t stream)
{
   // Measured on MI355X (LJ 80^3): a workgroup of the 3 live waves per cell runs the kernel in 3.94 ms, cap/64 = 4 waves per cell (the
   // tail wave exits at once) in 4.72 ms, single-wave workgroups in 5.64 ms.
   // waves per cell: sized to the occupancy the host last saw (+16 atoms of slack), never more than cap/64.  Cells that outgrow the
   // estimate stay correct (their waves take extra chunks).  COMD_LJ_WAVES=k forces k (tests use 1 to exercise the extra-chunk path).
   int w = sim->maxAtoms / 64;
   if (sim->max_atoms_cell > 0 && (sim->max_atoms_cell + 16 + 63) / 64 < w) w = (sim->max_atoms_cell + 16 + 63) / 64;
   { const char* e = getenv("COMD_LJ_WAVES"); if (e && atoi(e) > 0 && atoi(e) < w) w = atoi(e); }
   const int wavesPerBlock = w <= 4 ? w : 4;
   const unsigned nBlocks = w <= 4 ? (unsigned)num_cells : (unsigned)ceilDiv((long)num_cells * w, 4);
   // candidate lists of the waves (lj_kernels.h): rows sized for 70 % of the fullest stencil the host has seen -- the part of 27 cells
   // within the cutoff of a whole cell's box is 76 %, of a 64-atom slab of it 61 % -- and a wave whose row is too short walks the stencil.
   // COMD_LJ_PRUNE=0 switches the lists off (A/B measurements); COMD_LJ_LIST_CAP=n forces rows of n entries (tests: the fallback).
   const bool pruneEnv = !(getenv("COMD_LJ_PRUNE") && atoi(getenv("COMD_LJ_PRUNE")) == 0);
   LjPotentialGpu* lj = &sim->lj_pot;
   if (pruneEnv && !lj->waveCand && lj->packedCap == 0) {
      // the fullest cell right now (one blocking read, once per simulation; SimGpu.max_atoms_cell may not have been filled in yet)
      std::vector<int> counts((size_t)sim->boxes.nTotalBoxes);
      HIP_CHECK(hipMemcpyAsync(counts.data(), sim->boxes.nAtoms, counts.size() * sizeof(int), hipMemcpyDeviceToHost, S(stream)));
      HIP_CHECK(hipStreamSynchronize(S(stream)));
      int fullest = sim->max_atoms_cell;
      for (int c : counts) if (c > fullest) fullest = c;
      const int occ = fullest + 16 < sim->maxAtoms ? fullest + 16 : sim->maxAtoms;
      lj->waveCandWaves = w;
      lj->waveCandCap = ((int)(0.70 * 27 * occ) + 7) & ~7;
      { const char* e = getenv("COMD_LJ_LIST_CAP"); if (e && atoi(e) > 0) lj->waveCandCap = (atoi(e) + 7) & ~7; }
      lj->packedCap = ((occ + 7) & ~7) < sim->maxAtoms ? ((occ + 7) & ~7) : sim->maxAtoms;     // a stencil with a fuller cell falls back to the walk
      // list entries are 32-bit byte offsets into the packed records
      // ~250 B of list per atom (18 GB at 256^3) + the packed records (2.9 GB each): when that does not fit what the device has free (keeping 2 GB
      // for everything allocated later), or the 32-bit offsets cannot reach the records, this simulation walks the stencil as round 1 did
      const double listBytes = (double)sim->boxes.nLocalBoxes * lj->waveCandWaves * lj->waveCandCap * 4.0 + (double)sim->boxes.nLocalBoxes * lj->waveCandWaves * 8.0
                               + (sim->interior_stream ? 2.0 : 1.0) * (double)sim->boxes.nTotalBoxes * lj->packedCap * (sizeof(LjPos4) + sizeof(float4));
      size_t freeB = 0, totalB = 0;
      HIP_CHECK(hipMemGetInfo(&freeB, &totalB));
      if ((double)sim->boxes.nTotalBoxes * lj->packedCap * sizeof(LjPos4) >= 4294967296.0) lj->packedCap = -1;     // no lists for this simulation
      else if (listBytes + 2.0e9 > (double)freeB || (getenv("COMD_LJ_LIST_BUDGET_MB") && listBytes > 1.0e6 * atof(getenv("COMD_LJ_LIST_BUDGET_MB")))) {
         fprintf(stderr, "Rank %d: LJ candidate lists need %.1f GB, %.1f GB are free: running without them (the plain 27-cell walk, ~1.4x slower)\n",
                 g_rank, listBytes / 1e9, (double)freeB / 1e9);
         lj->packedCap = -1;
      } else {
         // Everything the lists need is allocated HERE, and a refusal is not fatal: ranks that share a device can all see the memory free and then not all
         // get it -- whoever is refused runs the plain walk, and says so (bench.py records force_path.lj_candidate_lists_active).
         auto tryAlloc = [](size_t bytes) { void* p = nullptr; if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { (void)hipGetLastError(); p = nullptr; } return p; };
         const size_t recs = (size_t)sim->boxes.nTotalBoxes * lj->packedCap * 4;
         lj->waveCand = (unsigned*)tryAlloc((size_t)sim->boxes.nLocalBoxes * lj->waveCandWaves * lj->waveCandCap * sizeof(unsigned));
         lj->waveCandCount = (int*)tryAlloc((size_t)sim->boxes.nLocalBoxes * lj->waveCandWaves * 2 * sizeof(int));
         bool ok = lj->waveCand && lj->waveCandCount;
         for (int k = 0; k < (sim->interior_stream ? 2 : 1) && ok; ++k) {
            lj->packedR[k] = (real_t*)tryAlloc(recs * sizeof(real_t));
            lj->packedF[k] = (float*)tryAlloc(recs * sizeof(float));
            ok = lj->packedR[k] && lj->packedF[k];
         }
         if (!ok) {
            void* ps[6] = { lj->waveCand, lj->waveCandCount, lj->packedR[0], lj->packedF[0], lj->packedR[1], lj->packedF[1] };
            for (void* q : ps) if (q) (void)hipFree(q);
            lj->waveCand = nullptr; lj->waveCandCount = nullptr; lj->packedR[0] = lj->packedR[1] = nullptr; lj->packedF[0] = lj->packedF[1] = nullptr;
            fprintf(stderr, "Rank %d: the device refused the %.1f GB of the LJ candidate lists: running without them (the plain 27-cell walk, ~1.4x slower)\n", g_rank, listBytes / 1e9);
            lj->packedCap = -1;
         }
      }
   }
   const bool prune = pruneEnv && lj->waveCand;
   LjWaveLists wl; memset(&wl, 0, sizeof wl);
   if (prune) {
      // the force may be split over two streams (-a 1: interior cells while the halo exchange is in flight, boundary cells after it):
      // each stream packs the positions it is about to read into its own array
      const int which = (cells_list && stream != sim->interior_stream) ? 1 : 0;
      if (!lj->packedR[which]) {
         lj->packedR[which] = dalloc<real_t>((size_t)sim->boxes.nTotalBoxes * lj->packedCap * 4, false);
         lj->packedF[which] = dalloc<float>((size_t)sim->boxes.nTotalBoxes * lj->packedCap * 4, false);
      }
      wl.cand = lj->waveCand; wl.pos = (const LjPos4*)lj->packedR[which]; wl.posF = (const float4*)lj->packedF[which]; wl.count = (int2*)lj->waveCandCount;
      wl.candCap = lj->waveCandCap; wl.wavesMax = lj->waveCandWaves; wl.capP = lj->packedCap;
      ljBoxMarginsF(sim, a.rc2, &wl.rc2BoxF, &wl.growF);
      // interior cells never have a halo cell in their stencil, and the halo cells are being filled while they run
      const int packCells = (cells_list && stream == sim->interior_stream) ? sim->boxes.nLocalBoxes : sim->boxes.nTotalBoxes;
      ForceTimer aux(sim, S(stream), 1);               // the list build of this evaluation (bench.py: force_evaluation_ms = kernel + this)
      hipLaunchKernelGGL(LJ_PackPositions, dim3((unsigned)ceilDiv((long)packCells * lj->packedCap, 256)), dim3(256), 0, S(stream),
                         a.rx, a.ry, a.rz, a.nAtoms, (LjPos4*)lj->packedR[which], (float4*)lj->packedF[which], a.cap, lj->packedCap, packCells, a.rc2,
                         sim->boxes.localMin[0], sim->boxes.localMin[1], sim->boxes.localMin[2]);
      hipLaunchKernelGGL(LJ_WaveCandidates, dim3((unsigned)ceilDiv(num_cells, 4)), dim3(256), 0, S(stream), a, wl, w);
   }
   ForceTimer timer(sim, S(stream));                     // the force kernel proper (bench.py's roofline line; rocprof must agree with it)
   if (prune) {
      if (sim->needEnergy) hipLaunchKernelGGL((LJ_Force_thread_atom<true, true>), dim3(nBlocks), dim3(64 * wavesPerBlock), 0, S(stream), a, w, wl);
      else              hipLaunchKernelGGL((LJ_Force_thread_atom<false, true>), dim3(nBlocks), dim3(64 * wavesPerBlock), 0, S(stream), a, w, wl);
   } else {
      if (sim->needEnergy) hipLaunchKernelGGL((LJ_Force_thread_atom<true, false>), dim3(nBlocks), dim3(64 * wavesPerBlock), 0, S(stream), a, w, wl);
      else              hipLaunchKernelGGL((LJ_Force_thread_atom<false, false>), dim3(nBlocks), dim3(64 * wavesPerBlock), 0, S(stream), a, w, wl);
   }
   LAUNCH_CHECK();
}

extern "C" void ljForceGpuAsync(SimGpu* sim, int num_cells, int* cells_list, int method, comdStream_t stream)
{
   if (num_cells <= 0) return;
   LjArgs a = makeLjArgs(sim, num_cells, cells_list);
   if (method != THREAD_ATOM_NL && method != WARP_ATOM_NL && method != CTA_CELL) { launchLjThreadAtom(sim, a, num_cells, cells_list, stream); return; }
   ForceTimer timer(sim, S(stream));
   if ((method == THREAD_ATOM_NL || method == WARP_ATOM_NL) && sim->atoms.neighborList.slabFormat) {
      NeighborListGpu* n = &sim->atoms.neighborList;
      (void)nlView(sim);
      NlSlabView v; v.list = n->list16; v.count = n->nNeighbors; v.rows = n->slabRows;
      const int threads = ((n->maxCellAtoms + 63) / 64) * 64;
      const size_t lds = (size_t)3 * n->maxSlabAtoms * sizeof(real_t);
      allowDynamicLds((const void*)LJ_Force_nl_slabs<true>, lds);
      allowDynamicLds((const void*)LJ_Force_nl_slabs<false>, lds);
      if (sim->needEnergy) hipLaunchKernelGGL(LJ_Force_nl_slabs<true>, dim3(num_cells), dim3(threads), lds, S(stream), a, v, n->maxSlabAtoms);
      else              hipLaunchKernelGGL(LJ_Force_nl_slabs<false>, dim3(num_cells), dim3(threads), lds, S(stream), a, v, n->maxSlabAtoms);
   } else if (method == THREAD_ATOM_NL || method == WARP_ATOM_NL) {
      const NlView nl = nlView(sim);
      const unsigned nBlocks = (unsigned)ceilDiv((long)num_cells * sim->maxAtoms, 256);
      if (sim->needEnergy) hipLaunchKernelGGL(LJ_Force_thread_atom_nl<true>, dim3(nBlocks), dim3(256), 0, S(stream), a, nl);
      else              hipLaunchKernelGGL(LJ_Force_thread_atom_nl<false>, dim3(nBlocks), dim3(256), 0, S(stream), a, nl);
   } else if (method == CTA_CELL) {
      const size_t lds = ljCtaLdsBytes(sim->maxAtoms);
      allowDynamicLds((const void*)LJ_Force_cta_cell<0, true>, lds); allowDynamicLds((const void*)LJ_Force_cta_cell<0, false>, lds);
      allowDynamicLds((const void*)LJ_Force_cta_cell<1, true>, lds); allowDynamicLds((const void*)LJ_Force_cta_cell<1, false>, lds);
      allowDynamicLds((const void*)LJ_Force_cta_cell<2, true>, lds); allowDynamicLds((const void*)LJ_Force_cta_cell<2, false>, lds);
      // threads = atoms of the fullest cell the host has seen (+16), rounded to whole waves, at most 256; each thread can own two atoms
      int threads = sim->maxAtoms < 256 ? sim->maxAtoms : 256;
      // (not with pairlists: their bits are per wave, so the thread -> atom map must not change between the generating call and the users)
      if (sim->atoms.neighborList.slabFormat != 3 && sim->max_atoms_cell > 0 && ((sim->max_atoms_cell + 16 + 63) / 64) * 64 < threads)
         threads = ((sim->max_atoms_cell + 16 + 63) / 64) * 64;
      if (sim->maxAtoms > 2 * threads) { fprintf(stderr, "ljForceGpu: cta_cell supports at most 512 atoms per cell\n"); exit(-1); }
      NeighborListGpu* n = &sim->atoms.neighborList;
      LjPairlist pl; pl.words = n->pairlist; pl.wavesMax = n->pairlistWaves;
      pl.plCut2 = (sim->lj_pot.cutoff + n->skinDistance) * (sim->lj_pot.cutoff + n->skinDistance);
#define LAUNCH_CTA(PLV) do { if (sim->needEnergy) hipLaunchKernelGGL((LJ_Force_cta_cell<PLV, true>), dim3(num_cells), dim3(threads), lds, S(stream), a, sim->status, pl); \
                            else              hipLaunchKernelGGL((LJ_Force_cta_cell<PLV, false>), dim3(num_cells), dim3(threads), lds, S(stream), a, sim->status, pl); } while (0)
      if (n->slabFormat != 3 && !(getenv("COMD_LJ_CTA_SLABS") && atoi(getenv("COMD_LJ_CTA_SLABS")) != 0)) {
         // the default form: every wave stages its own box-pruned candidates (COMD_LJ_CTA_SLABS=1: the slab kernel, for A/B runs)
         real_t rc2Box, grow;
         ljBoxMargins(sim, a.rc2, &rc2Box, &grow);
         const size_t ldsB = ljCtaBoxesLdsBytes(threads);
         if (sim->needEnergy) hipLaunchKernelGGL(LJ_Force_cta_cell_boxes<true>, dim3(num_cells), dim3(threads), ldsB, S(stream), a, rc2Box, grow);
         else                 hipLaunchKernelGGL(LJ_Force_cta_cell_boxes<false>, dim3(num_cells), dim3(threads), ldsB, S(stream), a, rc2Box, grow);
      }
      else if (n->slabFormat != 3)               LAUNCH_CTA(0);
      else if (n->nBuilds == 0) { fprintf(stderr, "ljForceGpu: -L needs buildNeighborListGpu before the first force call\n"); exit(-1); }
      else if (n->pairlistBuildId != n->nBuilds) LAUNCH_CTA(1);
      else                                       LAUNCH_CTA(2);
#undef LAUNCH_CTA
   }
   LAUNCH_CHECK();
}

extern "C" void ljForceGpu(SimGpu* sim, int interpolation, int num_cells, int* cells_list, real_t plcutoff, int method)
{
   (void)plcutoff;
   if (interpolation != 0) { fprintf(stderr, "ljForceGpu: table-interpolated LJ (-I) is out of scope\n"); exit(-1); }
   ljForceGpuAsync(sim, num_cells, cells_list, method, nullptr);
}

static EamArgs makeEamArgs(SimGpu* sim, int num_cells, int* cells_list)
{
   EamArgs a;
   a.rx = sim->atoms.r.x; a.ry = sim->atoms.r.y; a.rz = sim->atoms.r.z;
   a.fx = sim->atoms.f.x; a.fy = sim->atoms.f.y; a.fz = sim->atoms.f.z; a.e = sim->atoms.e;
   a.rhobar = sim->eam_pot.rhobar; a.dfEmbed = sim->eam_pot.dfEmbed;
   a.nAtoms = sim->boxes.nAtoms; a.nbr = sim->neighbor_cells; a.cells = cells_list;
   a.nCells = num_cells; a.cap = sim->maxAtoms;
   a.rc2 = sim->eam_pot.cutoff * sim->eam_pot.cutoff;
   a.phi = sim->eam_pot.phi; a.rho = sim->eam_pot.rho; a.f = sim->eam_pot.f;
   a.phiS = sim->eam_pot.phiS; a.rhoS = sim->eam_pot.rhoS;
   a.sel = nullptr; a.tag = 0;
   return a;
}

// thread per atom: lanes per cell = the fullest cell the host has seen (+ 2), as a power of two; persistent workgroups when the tables
// sit in the LDS (8 per CU's worth of 256 CUs), one workgroup per 256 / lanesPerCell cells otherwise
template <int STEP>
static void launchEamThreadAtom(SimGpu* sim, const EamArgs& a, int num_cells, hipStream_t st, bool spline)
{
   int want = sim->max_atoms_cell > 0 ? sim->max_atoms_cell + 2 : sim->maxAtoms;
   if (want > sim->maxAtoms) want = sim->maxAtoms;
   int lanes = 4;
   while (lanes < want && lanes < 256) lanes *= 2;
   const int nGroups = ceilDiv(num_cells, 256 / lanes);
   const size_t tableBytes = eamCtaTableBytes(STEP, a.rho.n, a.phi.n);
   if (spline) {
      hipLaunchKernelGGL((EAM_Force_thread_atom<STEP, true, false>), dim3(nGroups), dim3(256), 0, st, a, lanes);
   } else if (tableBytes <= 32 * 1024) {
      const int grid = nGroups < 4096 ? nGroups : 4096;
      hipLaunchKernelGGL((EAM_Force_thread_atom<STEP, false, true>), dim3(grid), dim3(256), tableBytes, st, a, lanes);
   } else {
      hipLaunchKernelGGL((EAM_Force_thread_atom<STEP, false, false>), dim3(nGroups), dim3(256), 0, st, a, lanes);
   }
   LAUNCH_CHECK();
}

// cta_cell, brick form (eam_brick_kernels.h): a workgroup stages the cells around a brick of 1 x BY x BZ cells once and its waves take the
// brick's cells one at a time.  COMD_EAM_BRICK="by,bz" overrides the brick (experiments).  The Verlet-list method of EAM (slabFormat 4) runs on the
// same kernel with LISTED = true: rows built once per list build, both passes read them back.
static bool eamListedBrick(const SimGpu* sim, int method)
{
   return (method == THREAD_ATOM_NL || method == WARP_ATOM_NL) && sim->atoms.neighborList.slabFormat == 4;
}
static bool eamBrickPath(const SimGpu* sim, int method)
{
   if (eamListedBrick(sim, method)) return true;
   return method == CTA_CELL && !(getenv("COMD_EAM_CTA") && !strcmp(getenv("COMD_EAM_CTA"), "cell"))
          && (double)sim->boxes.nTotalBoxes * sim->maxAtoms * sizeof(real_t) < 4294967296.0;      // (the brick kernel stages with 32-bit byte offsets)
}

// thread_atom on the brick image (eam_atom_brick_kernels.h).  COMD_EAM_THREAD_ATOM=cell keeps round 2's kernel (a share of a wave per cell, candidates streamed
// through L2; A/B runs), as do arrays of 4 GiB or more (the staging uses 32-bit byte offsets).
static bool eamAtomBrickPath(const SimGpu* sim, int method)
{
   return (method == THREAD_ATOM || method == WARP_ATOM) && !(getenv("COMD_EAM_THREAD_ATOM") && !strcmp(getenv("COMD_EAM_THREAD_ATOM"), "cell"))
          && (double)sim->boxes.nTotalBoxes * sim->maxAtoms * sizeof(real_t) < 4294967296.0;
}

// The overlap mode's two lists as brick groups (eam_brick_kernels.h ClassifyBrickCells): 1 = this is the launch over SimGpu.boundary_cells, 2 = over
// SimGpu.interior_cells, 0 = any other list (cell marks).  COMD_EAM_GROUPS=0 keeps the lists as they are given (A/B runs, tests).
static int eamBrickGroupOf(const SimGpu* sim, const int* cells_list, int num_cells, int method)
{
   if (!cells_list || !(eamBrickPath(sim, method) || eamAtomBrickPath(sim, method))) return 0;
   if (getenv("COMD_EAM_GROUPS") && atoi(getenv("COMD_EAM_GROUPS")) == 0) return 0;
   if (cells_list == sim->boundary_cells && num_cells == sim->n_boundary_cells) return 1;
   if (cells_list == sim->interior_cells && num_cells == sim->n_interior_cells) return 2;
   return 0;
}

// The brick shape of a simulation: 1 x 4 x 2 cells unless COMD_EAM_BRICK says otherwise; fixed by the first launch (rows index the image of that shape).
static void eamBrickShape(SimGpu* sim, bool listed, int* by, int* bz)
{
   if (!sim->eam_pot.brickBy) {
      int y = 4, z = 2;
      const int maxCells = listed ? (EAM_BRICK_STAGE_LISTED * 256) / 32 : EAM_BRICK_MAX_CELLS;      // what the staging loop covers (eam_brick_kernels.h)
      const char* e = getenv("COMD_EAM_BRICK"); int ey = 0, ez = 0;
      if (e && sscanf(e, "%d,%d", &ey, &ez) == 2 && ey >= 1 && ez >= 1 && 3 * (ey + 2) * (ez + 2) <= maxCells && ey * ez <= 64) { y = ey; z = ez; }
      sim->eam_pot.brickBy = y; sim->eam_pot.brickBz = z;
   }
   *by = sim->eam_pot.brickBy; *bz = sim->eam_pot.brickBz;
}

static void eamBrickGeometry(SimGpu* sim, bool listed, EamBrickArgs* b)
{
   memset(b, 0, sizeof *b);
   for (int k = 0; k < 3; ++k) { b->geom.g[k] = sim->boxes.gridSize[k]; b->geom.lmin[k] = sim->boxes.localMin[k]; b->geom.lmax[k] = sim->boxes.localMax[k]; b->geom.inv[k] = sim->boxes.invBoxSize[k]; }
   b->geom.nLocal = sim->boxes.nLocalBoxes; b->geom.nTotal = sim->boxes.nTotalBoxes;
   b->geom.lookup = sim->boxes.boxIDLookUp; b->geom.reverse = sim->boxes.boxIDLookUpReverse;
   eamBrickShape(sim, listed, &b->by, &b->bz);
   b->nby = ceilDiv(b->geom.g[1], b->by); b->nbz = ceilDiv(b->geom.g[2], b->bz);
}

// The image must hold the atoms of the fullest BLOCK (3 x (by + 2) x (bz + 2) cells), not the mean: the lattice and the cell grid are incommensurate,
// and at 80^3 the blocks of a 1 x 4 x 2 brick hold 755 atoms on average and up to 918.  A brick whose block outgrows the image takes the
// thread-per-atom form (correct, many times slower), so the occupancies are read once, the fullest block of this brick shape is found and the image
// sized for it + 1 % + 8 (blocks gain or lose a handful of atoms through their surface as the lattice moves).  Both passes use that size.
// Called by the first launch; by every Verlet-list build (the cells were just re-binned); and again when comdEamBrickStats finds bricks in the fall-back.
static int eamBrickSizeImage(SimGpu* sim, const EamBrickArgs& b, hipStream_t st, bool listed)
{
   const double cellVol = 1.0 / (sim->boxes.invBoxSize[0] * sim->boxes.invBoxSize[1] * sim->boxes.invBoxSize[2]);
   const double lat = sim->latticeConstant > 0.0 ? sim->latticeConstant : 3.615;
   std::vector<int> counts((size_t)sim->boxes.nTotalBoxes), lookup;
   HIP_CHECK(hipMemcpyAsync(counts.data(), sim->boxes.nAtoms, counts.size() * sizeof(int), hipMemcpyDeviceToHost, st));
   if (sim->boxes.boxIDLookUp) {
      lookup.resize((size_t)sim->boxes.nLocalBoxes);
      HIP_CHECK(hipMemcpyAsync(lookup.data(), sim->boxes.boxIDLookUp, lookup.size() * sizeof(int), hipMemcpyDeviceToHost, st));
   }
   HIP_CHECK(hipStreamSynchronize(st));
   CellGeom hg = b.geom; hg.lookup = lookup.empty() ? nullptr : lookup.data(); hg.reverse = nullptr;
   const int gx = hg.g[0], gy = hg.g[1], gz = hg.g[2];
   // per (y, z) row of three x cells, then the (by + 2) x (bz + 2) window of rows around every brick
   long fullest = 0;
   std::vector<int> row3((size_t)gx * (gy + 2) * (gz + 2));
   for (int z = -1; z <= gz; ++z) for (int y = -1; y <= gy; ++y) for (int x = 0; x < gx; ++x)
      row3[(size_t)x + (size_t)gx * ((y + 1) + (size_t)(gy + 2) * (z + 1))] =
         counts[comdBoxFromTuple(&hg, x - 1, y, z)] + counts[comdBoxFromTuple(&hg, x, y, z)] + counts[comdBoxFromTuple(&hg, x + 1, y, z)];
   // (only blocks made of local cells count: with -a 1 the first launch runs while the halo cells are still being filled; the lattice is periodic, the
   // blocks at the faces are no fuller than those inside.  A grid too small to have such blocks takes the mean density + 25 %.)
   for (int bzI = 0; bzI < b.nbz; ++bzI) for (int byI = 0; byI < b.nby; ++byI) for (int x = 1; x < gx - 1; ++x) {
      if (byI * b.by - 1 < 0 || byI * b.by + b.by > gy - 1 || bzI * b.bz - 1 < 0 || bzI * b.bz + b.bz > gz - 1) continue;
      long sum = 0;
      for (int z = bzI * b.bz - 1; z <= bzI * b.bz + b.bz; ++z)
         for (int y = byI * b.by - 1; y <= byI * b.by + b.by; ++y) sum += row3[(size_t)x + (size_t)gx * ((y + 1) + (size_t)(gy + 2) * (z + 1))];
      if (sum > fullest) fullest = sum;
   }
   if (fullest == 0) fullest = (long)(1.25 * 3 * (b.by + 2) * (b.bz + 2) * cellVol * 4.0 / (lat * lat * lat));
   int cap = (((int)(fullest * 1.01) + 8 + (listed ? 1 : 0) + 7) / 8) * 8;      // (listed launches keep one more record: the far-away one that pads odd rows)
   if (cap < 256) cap = 256;
   if (cap > 4096) cap = 4096;                            // 16-bit numbers would reach 65535; beyond 4096 records the cells take the thread-per-atom form
   { const char* e = getenv("COMD_EAM_IMAGE"); if (e && atoi(e) >= 64) cap = (atoi(e) + 7) / 8 * 8; }      // experiments / tests: force the fallback
   return cap;
}

// Verlet rows: the brick lists of a list build.  The occupancies are final (the atom exchange has run) and frozen until the next build, so the host can
// look at every block once: the image is sized for what the passes can keep four workgroups per CU with, a brick whose block would outgrow it is listed as
// its two z halves (eam_brick_kernels.h), and the boundary / interior launches of the overlap mode get their lists of whole bricks here as well.
static size_t eamBrickTableDoubles(const SimGpu* sim, int step, int spline)
{
   const EamPotentialGpu& e = sim->eam_pot;
   if (step == 0 || spline || eamCtaTableBytes(step, e.rho.n, e.phi.n) > 32 * 1024) return 0;
   const bool sameGrid = e.phi.n == e.rho.n && e.phi.x0 == e.rho.x0 && e.phi.invDx == e.rho.invDx;
   return step == 1 ? (size_t)2 * (e.rho.n + 3) + (sameGrid ? 0 : (e.phi.n + 3 - (e.rho.n + 3))) : (size_t)(e.rho.n + 3);
}

static void eamBrickBuildLists(SimGpu* sim, hipStream_t st, int spline)
{
   NeighborListGpu* n = &sim->atoms.neighborList;
   EamBrickArgs b;
   eamBrickGeometry(sim, true, &b);
   const int gx = b.geom.g[0], gy = b.geom.g[1], gz = b.geom.g[2], nBricks = gx * b.nby * b.nbz;
   std::vector<int> counts((size_t)sim->boxes.nTotalBoxes), lookup, boundary((size_t)(sim->boundary_cells ? sim->n_boundary_cells : 0));
   HIP_CHECK(hipMemcpyAsync(counts.data(), sim->boxes.nAtoms, counts.size() * sizeof(int), hipMemcpyDeviceToHost, st));
   if (sim->boxes.boxIDLookUp) {
      lookup.resize((size_t)sim->boxes.nLocalBoxes);
      HIP_CHECK(hipMemcpyAsync(lookup.data(), sim->boxes.boxIDLookUp, lookup.size() * sizeof(int), hipMemcpyDeviceToHost, st));
   }
   if (!boundary.empty()) HIP_CHECK(hipMemcpyAsync(boundary.data(), sim->boundary_cells, boundary.size() * sizeof(int), hipMemcpyDeviceToHost, st));
   HIP_CHECK(hipStreamSynchronize(st));
   CellGeom hg = b.geom; hg.lookup = lookup.empty() ? nullptr : lookup.data(); hg.reverse = nullptr;
   // atoms of the three x cells around (x, y, z), y and z from -1 to g
   std::vector<int> row3((size_t)gx * (gy + 2) * (gz + 2));
   auto r3 = [&](int x, int y, int z) -> int& { return row3[(size_t)x + (size_t)gx * ((y + 1) + (size_t)(gy + 2) * (z + 1))]; };
   for (int z = -1; z <= gz; ++z) for (int y = -1; y <= gy; ++y) for (int x = 0; x < gx; ++x)
      r3(x, y, z) = counts[comdBoxFromTuple(&hg, x - 1, y, z)] + counts[comdBoxFromTuple(&hg, x, y, z)] + counts[comdBoxFromTuple(&hg, x + 1, y, z)];
   // records in the image of the brick part [z0, z0 + nz) of brick (x, byI, bzI): the kernel stages rows y0-1 .. y0+by and planes z0-1 .. z0+nz that lie inside -1 .. g
   auto blockAtoms = [&](int x, int byI, int z0, int nz) {
      long sum = 0;
      for (int z = z0 - 1; z <= z0 + nz && z <= gz; ++z)
         for (int y = byI * b.by - 1; y <= byI * b.by + b.by && y <= gy; ++y) sum += r3(x, y, z);
      return sum;
   };
   std::vector<long> whole((size_t)nBricks);
   long fullest = 0;
   for (int i = 0; i < nBricks; ++i) {
      const int x = i % gx, byI = (i / gx) % b.nby, bzI = i / (gx * b.nby);
      whole[i] = blockAtoms(x, byI, bzI * b.bz, b.bz);
      if (whole[i] > fullest) fullest = whole[i];
   }
   // the largest image that leaves four workgroups per CU (160 KB of LDS in 1280-byte granules) in pass 1 and in pass 3
   const int waves = 4;
   auto perCu = [&](int step, int cap) {
      const size_t lds = eamBrickLdsBytes(step, true, eamBrickTableDoubles(sim, step, spline), cap, n->brickRowLen, waves);
      return lds > 160 * 1024 ? 0 : (int)(160 * 1024 / (((lds + 1279) / 1280) * 1280));
   };
   int capFull = (((int)fullest + 1 + 7) / 8) * 8;            // (+ 1: the far-away record; nothing moves between builds, so no head-room)
   if (capFull < 256) capFull = 256;
   int cap = capFull;
   if (cap <= 4096 && (perCu(1, cap) < 4 || perCu(3, cap) < 4) && b.bz % 2 == 0) {
      int fit = cap;
      while (fit > 256 && (perCu(1, fit) < 4 || perCu(3, fit) < 4)) fit -= 8;
      long split = 0;
      for (int i = 0; i < nBricks; ++i) split += whole[i] + 1 > fit;
      if (split * 10 <= nBricks) cap = fit;                  // worth it while at most one brick in ten is staged twice
   }
   if (cap > 4096) cap = 4096;
   { const char* e = getenv("COMD_EAM_IMAGE"); if (e && atoi(e) >= 64) cap = (atoi(e) + 7) / 8 * 8; }      // experiments / tests: force halves and the fall-back
   sim->eam_pot.brickImageCap = cap;
   const int headroom = cap / 32 > 8 ? cap / 32 : 8;
   // the lists: [0, stride) bricks that hold a boundary cell, [stride, 2 stride) the others, [2 stride, 3 stride) all of them; brick order, halves adjacent
   std::vector<char> isBoundary((size_t)sim->boxes.nLocalBoxes, 0);
   for (int c : boundary) if (c >= 0 && c < sim->boxes.nLocalBoxes) isBoundary[c] = 1;
   const int stride = 2 * nBricks;
   std::vector<int> lists((size_t)3 * stride, 0), group((size_t)sim->boxes.nLocalBoxes, 2);
   int cnt[3] = { 0, 0, 0 };
   for (int i = 0; i < nBricks; ++i) {
      const int x = i % gx, by0 = ((i / gx) % b.nby) * b.by, bz0 = (i / (gx * b.nby)) * b.bz;
      bool any = false;
      for (int dz = 0; dz < b.bz; ++dz) for (int dy = 0; dy < b.by; ++dy)
         if (by0 + dy < gy && bz0 + dz < gz) any = any || isBoundary[comdBoxFromTuple(&hg, x, by0 + dy, bz0 + dz)];
      for (int dz = 0; dz < b.bz; ++dz) for (int dy = 0; dy < b.by; ++dy)
         if (by0 + dy < gy && bz0 + dz < gz) group[comdBoxFromTuple(&hg, x, by0 + dy, bz0 + dz)] = any ? 1 : 2;
      const int g = any ? 0 : 1;
      // (a brick within 3 % of the image goes in halves too: the lists outlive this build -- atoms wander between cells from one build to the next -- and a brick
      // that outgrows the image later takes the thread-per-atom form until the lists are made again)
      const bool halves = whole[i] + 1 > cap - headroom && b.bz % 2 == 0 && bz0 + b.bz / 2 < gz;      // (an upper half outside the grid would be an empty workgroup)
      const int e[2] = { halves ? i | (1 << 28) : i, i | (2 << 28) };
      for (int k = 0; k < (halves ? 2 : 1); ++k) { lists[(size_t)g * stride + cnt[g]++] = e[k]; lists[(size_t)2 * stride + cnt[2]++] = e[k]; }
   }
   if (sim->eam_pot.brickList && sim->eam_pot.brickListStride != stride) { HIP_CHECK(hipFree(sim->eam_pot.brickList)); sim->eam_pot.brickList = nullptr; }
   if (!sim->eam_pot.brickList) sim->eam_pot.brickList = dalloc<int>((size_t)3 * stride, false);
   if (!sim->eam_pot.brickGroup) sim->eam_pot.brickGroup = dalloc<int>((size_t)sim->boxes.nLocalBoxes, false);
   HIP_CHECK(hipMemcpyAsync(sim->eam_pot.brickList, lists.data(), lists.size() * sizeof(int), hipMemcpyHostToDevice, st));
   HIP_CHECK(hipMemcpyAsync(sim->eam_pot.brickGroup, group.data(), group.size() * sizeof(int), hipMemcpyHostToDevice, st));
   HIP_CHECK(hipStreamSynchronize(st));                      // (the vectors go out of scope; the other stream of the overlap mode reads the lists too)
   sim->eam_pot.brickCount[0] = cnt[0]; sim->eam_pot.brickCount[1] = cnt[1]; sim->eam_pot.brickCountAll = cnt[2]; sim->eam_pot.brickListStride = stride;
   sim->eam_pot.brickGroupBy = b.by; sim->eam_pot.brickGroupBz = b.bz;
   sim->eam_pot.brickListMakes++;
}

// The bricks of the boundary and of the interior launch as lists, for the brick shape in `b` (built once per shape; eam_pot.brickGroup marks the cells for
// kernels over cells).  Shared by cta_cell and thread_atom on the brick image.
static void eamBrickGroupLists(SimGpu* sim, const EamBrickArgs& b, hipStream_t st)
{
   if (sim->eam_pot.brickGroup && sim->eam_pot.brickGroupBy == b.by && sim->eam_pot.brickGroupBz == b.bz) return;
   if (!sim->eam_pot.brickGroup) sim->eam_pot.brickGroup = dalloc<int>((size_t)sim->boxes.nLocalBoxes, false);
   if (!sim->eam_pot.cellSel) {
      sim->eam_pot.cellSel = dalloc<int>((size_t)sim->boxes.nLocalBoxes, false);
      HIP_CHECK(hipMemsetAsync(sim->eam_pot.cellSel, 0, (size_t)sim->boxes.nLocalBoxes * sizeof(int), st));
   }
   const int tag = ++sim->eam_pot.selTag;
   const EamBrickArgs g = b;
   if (sim->n_boundary_cells > 0)
      hipLaunchKernelGGL(MarkCells, dim3(ceilDiv(sim->n_boundary_cells, 256)), dim3(256), 0, st, sim->boundary_cells, sim->n_boundary_cells, sim->eam_pot.cellSel, tag);
   const int nBricks = b.geom.g[0] * b.nby * b.nbz;
   if (sim->eam_pot.brickList) HIP_CHECK(hipFree(sim->eam_pot.brickList));
   sim->eam_pot.brickList = dalloc<int>((size_t)2 * nBricks, false);
   hipLaunchKernelGGL(ClassifyBrickCells, dim3(ceilDiv(nBricks, 256)), dim3(256), 0, st, g, sim->eam_pot.cellSel, tag, sim->eam_pot.brickGroup, sim->eam_pot.brickList);
   // the bricks of either group as a list (built once; the other stream of the overlap mode reads groups and lists too, so wait here):
   // [0, n1) the bricks that hold a boundary cell, [nBricks, nBricks + n2) the others, each in brick order
   std::vector<int> cls((size_t)nBricks), lists((size_t)2 * nBricks, 0);
   HIP_CHECK(hipMemcpyAsync(cls.data(), sim->eam_pot.brickList, (size_t)nBricks * sizeof(int), hipMemcpyDeviceToHost, st));
   HIP_CHECK(hipStreamSynchronize(st));
   int n1 = 0, n2 = 0;
   for (int i = 0; i < nBricks; ++i) { if (cls[i] == 1) lists[n1++] = i; else lists[(size_t)nBricks + n2++] = i; }
   HIP_CHECK(hipMemcpyAsync(sim->eam_pot.brickList, lists.data(), lists.size() * sizeof(int), hipMemcpyHostToDevice, st));
   HIP_CHECK(hipStreamSynchronize(st));
   sim->eam_pot.brickCount[0] = n1; sim->eam_pot.brickCount[1] = n2; sim->eam_pot.brickListStride = nBricks;
   sim->eam_pot.brickGroupBy = b.by; sim->eam_pot.brickGroupBz = b.bz;
}

template <int STEP>
static void launchEamBrick(SimGpu* sim, const EamArgs& a, int num_cells, int* cells_list, hipStream_t st, int spline, bool listed, int method)
{
   const size_t tableBytes = eamCtaTableBytes(STEP == 0 ? 1 : STEP, a.rho.n, a.phi.n);
   const bool tablesInLds = STEP != 0 && !spline && tableBytes <= 32 * 1024;      // funcfl tables (500 samples) live in the LDS; setfl (10000) and spline coefficients stay in L2
   const bool sameGrid = a.phi.n == a.rho.n && a.phi.x0 == a.rho.x0 && a.phi.invDx == a.rho.invDx;
   size_t tableDoubles = 0;
   if (tablesInLds) tableDoubles = STEP == 1 ? (size_t)2 * (a.rho.n + 3) + (sameGrid ? 0 : (a.phi.n + 3 - (a.rho.n + 3))) : (size_t)(a.rho.n + 3);
   EamBrickArgs b;
   eamBrickGeometry(sim, listed, &b);
   const double lat = sim->latticeConstant > 0.0 ? sim->latticeConstant : 3.615;
   if (!sim->eam_pot.brickImageCap) sim->eam_pot.brickImageCap = eamBrickSizeImage(sim, b, st, listed);
   b.imageCap = sim->eam_pot.brickImageCap;
   if (!sim->eam_pot.brickStats) sim->eam_pot.brickStats = dalloc<int>(2);
   b.stats = sim->eam_pot.brickStats;
   if (listed) {
      NeighborListGpu* n = &sim->atoms.neighborList;
      b.rows = n->brickRowLen; b.rowsG = n->brickRows; b.rowCountG = n->brickRowCount;
      b.listRounds = n->brickRounds; b.listQuads = n->brickQuads;
      const real_t rBuild = sim->eam_pot.cutoff + n->skinDistance;
      b.rBuild2 = rBuild * rBuild;
   } else {
      // rows per atom: the cutoff sphere at that density + 50 %, a multiple of 8
      const double rc = sim->eam_pot.cutoff;
      int rows = ((int)(4.18879020478639 * rc * rc * rc * 4.0 / (lat * lat * lat) * 1.5) + 7) / 8 * 8;
      if (rows < 32) rows = 32;
      if (rows > 256) rows = 256;
      const int lanesMin = (rows + 15) / 16, roundAtoms = 64 / lanesMin < 16 ? 64 / lanesMin : 16;      // (as the kernel derives them from `rows`)
      const int rounds = (sim->maxAtoms + roundAtoms - 1) / roundAtoms;
      if (!sim->eam_pot.pairRows) {                             // rows pass 1 leaves for pass 3: per (cell, round) [2 quads][64 lanes] 16-byte elements
         const size_t slotsLocal = (size_t)sim->boxes.nLocalBoxes * sim->maxAtoms;
         sim->eam_pot.pairRows = dalloc<unsigned>((size_t)sim->boxes.nLocalBoxes * rounds * 2 * 64 * 4, false);
         sim->eam_pot.pairRowCount = dalloc<unsigned short>(slotsLocal, false);      // (written by pass 1 before pass 3 reads it; no zeroing that could race with that)
         sim->eam_pot.pairRowLen = rows;
      }
      b.rows = sim->eam_pot.pairRowLen; b.rowsG = sim->eam_pot.pairRows; b.rowCountG = sim->eam_pot.pairRowCount;
      b.listRounds = rounds; b.listQuads = 2;
      if (!sim->eam_pot.brickSel) {      // (zeroed on the launch stream, like the cell marks below)
         sim->eam_pot.brickSel = dalloc<unsigned long long>((size_t)sim->boxes.nLocalBoxes, false);
         HIP_CHECK(hipMemsetAsync(sim->eam_pot.brickSel, 0, (size_t)sim->boxes.nLocalBoxes * sizeof(unsigned long long), st));
      }
      b.brickSel = sim->eam_pot.brickSel;
   }
   b.fuseEmbed = sim->fuseEmbed; b.status = sim->status;
   { const char* e = getenv("COMD_EAM_ABLATE"); b.debug = e ? atoi(e) : 0; }
   const int group = eamBrickGroupOf(sim, cells_list, num_cells, method);
   if (listed) {           // the lists of the last list build (eamBrickBuildLists): all bricks, or the whole bricks of the boundary / interior launch
      if (!sim->eam_pot.brickList) { fprintf(stderr, "eamForce: thread_atom_nl needs buildNeighborListGpu before the first force call\n"); exit(-1); }
      b.brickList = sim->eam_pot.brickList + (size_t)(group ? group - 1 : 2) * sim->eam_pot.brickListStride;
   }
   if (group && !listed) {            // the boundary / interior launch of the overlap mode: whole bricks (a brick with cells of both lists would be staged twice per pass)
      eamBrickGroupLists(sim, b, st);
      // every cell of a listed brick is selected: no marks to look at (the embedding pass, a kernel over cells, uses brickGroup)
      b.brickList = sim->eam_pot.brickList + (group == 1 ? 0 : sim->eam_pot.brickListStride);
   } else if (cells_list && !group) {      // a launch over any other cell list: mark the cells, every brick looks at its own
      if (!sim->eam_pot.cellSel) {
         // zeroed ON THE LAUNCH STREAM: hipMemset returns before the device has finished, and the -a 1 streams are non-blocking -- a zeroing on
         // the null stream can land after the marks of the first launch (seen once in four-rank runs: a first force evaluation that skipped cells)
         sim->eam_pot.cellSel = dalloc<int>((size_t)sim->boxes.nLocalBoxes, false);
         HIP_CHECK(hipMemsetAsync(sim->eam_pot.cellSel, 0, (size_t)sim->boxes.nLocalBoxes * sizeof(int), st));
      }
      b.sel = sim->eam_pot.cellSel; b.tag = ++sim->eam_pot.selTag;
      ForceTimer aux(sim, st, 1);
      hipLaunchKernelGGL(MarkCells, dim3(ceilDiv(num_cells, 256)), dim3(256), 0, st, cells_list, num_cells, sim->eam_pot.cellSel, b.tag);
   }
   const int waves = 4;     // EAM_Force_cta_brick is written for 256 threads: __launch_bounds__(256, 4), staging loops of STAGE x 256 tasks
   if (listed && 3 * (b.by + 2) * (b.bz + 2) * 32 > EAM_BRICK_STAGE_LISTED * 64 * waves) { fprintf(stderr, "eamForce: a brick of 1 x %d x %d cells has more cells around it than a listed launch stages\n", b.by, b.bz); exit(-1); }
   const size_t lds = eamBrickLdsBytes(STEP, listed, tableDoubles, b.imageCap, b.rows, waves);
   if (lds > 160 * 1024) { fprintf(stderr, "eamForce: cta_cell needs %zu bytes of LDS for this box\n", lds); exit(-1); }
   const int grid = group ? sim->eam_pot.brickCount[group - 1] : listed ? sim->eam_pot.brickCountAll : b.geom.g[0] * b.nby * b.nbz;
   if (grid <= 0) return;
   // the table clamps of interpolate() are dead weight when every pair the kernel evaluates lies inside the tables: 0 < r <= cutoff (rows hold pairs inside the cutoff;
   // listed pairs are evaluated at min(r, cutoff)), tables from x0 <= 0 up to xn >= cutoff.  COMD_EAM_CLAMP=1 keeps them (A/B runs).
   const double rcut = sim->eam_pot.cutoff * (1.0 + 4e-16);
   const bool clampFree = !spline && a.phi.x0 <= R(0.0) && a.rho.x0 <= R(0.0) && rcut <= (double)a.phi.xn && rcut <= (double)a.rho.xn
                          && !(getenv("COMD_EAM_CLAMP") && atoi(getenv("COMD_EAM_CLAMP")) != 0);
#define COMD_LAUNCH_EAM_BRICK(STP, TAB, SPL, LST, CLP) do { \
      allowDynamicLds((const void*)EAM_Force_cta_brick<STP, TAB, SPL, LST, CLP>, lds); \
      hipLaunchKernelGGL((EAM_Force_cta_brick<STP, TAB, SPL, LST, CLP>), dim3(grid), dim3(64 * waves), lds, st, a, b); } while (0)
#define COMD_LAUNCH_EAM_BRICK_C(STP, TAB, LST) do { if (clampFree) COMD_LAUNCH_EAM_BRICK(STP, TAB, false, LST, false); else COMD_LAUNCH_EAM_BRICK(STP, TAB, false, LST, true); } while (0)
   if (STEP == 0)        COMD_LAUNCH_EAM_BRICK(0, false, false, true, true);
   else if (listed) {
      if (spline)           COMD_LAUNCH_EAM_BRICK((STEP == 0 ? 1 : STEP), false, true, true, true);
      else if (tablesInLds) COMD_LAUNCH_EAM_BRICK_C((STEP == 0 ? 1 : STEP), true, true);
      else                  COMD_LAUNCH_EAM_BRICK_C((STEP == 0 ? 1 : STEP), false, true);
   } else {
      if (spline)           COMD_LAUNCH_EAM_BRICK((STEP == 0 ? 1 : STEP), false, true, false, true);
      else if (tablesInLds) COMD_LAUNCH_EAM_BRICK_C((STEP == 0 ? 1 : STEP), true, false);
      else                  COMD_LAUNCH_EAM_BRICK_C((STEP == 0 ? 1 : STEP), false, false);
   }
#undef COMD_LAUNCH_EAM_BRICK_C
#undef COMD_LAUNCH_EAM_BRICK
   LAUNCH_CHECK();
}

template <int STEP>
static void launchEamAtomBrick(SimGpu* sim, const EamArgs& a, int num_cells, int* cells_list, hipStream_t st, int spline)
{
   const size_t tableBytes = eamCtaTableBytes(STEP, a.rho.n, a.phi.n);
   const bool tablesInLds = !spline && tableBytes <= 32 * 1024;
   const bool sameGrid = a.phi.n == a.rho.n && a.phi.x0 == a.rho.x0 && a.phi.invDx == a.rho.invDx;
   auto tableDoublesOf = [&](int step) -> size_t {
      if (!tablesInLds) return 0;
      return step == 1 ? (size_t)2 * (a.rho.n + 3) + (sameGrid ? 0 : (a.phi.n + 3 - (a.rho.n + 3))) : (size_t)(a.rho.n + 3);
   };
   const double lat = sim->latticeConstant > 0.0 ? sim->latticeConstant : 3.615;
   const double rc = sim->eam_pot.cutoff;
   // a row per thread (bytes: a neighbour is its offset inside its run of the image): the cutoff sphere at the lattice's density + 50 %, a multiple of 8 (an atom
   // with more neighbours walks its stencil a second time); further down it gives up to a fifth of that when the LDS so freed buys a workgroup per CU in pass 1
   const double inSphere = 4.18879020478639 * rc * rc * rc * 4.0 / (lat * lat * lat);
   int rows = ((int)(inSphere * 1.5) + 7) / 8 * 8;
   if (rows < 32) rows = 32;
   if (rows > 128) rows = 128;
   int rowsForced = 0;
   { const char* e = getenv("COMD_EAM_ATOM_ROWS"); if (e && atoi(e) >= 8 && atoi(e) <= 128) rows = rowsForced = atoi(e) / 8 * 8; }      // tests: rows that overflow
   EamBrickArgs b;
   { const int by = sim->eam_pot.brickBy, bz = sim->eam_pot.brickBz; eamBrickGeometry(sim, false, &b); sim->eam_pot.brickBy = by; sim->eam_pot.brickBz = bz; }      // (the grid; the shape is this method's own)
   const double perCell = 4.0 / (lat * lat * lat) / (sim->boxes.invBoxSize[0] * sim->boxes.invBoxSize[1] * sim->boxes.invBoxSize[2]);      // atoms of a cell at the lattice's density
   // the threads that take atoms: whole waves for the brick's atoms + 8 % (a fuller brick's threads take a second atom)
   auto rowThreadsOf = [&](double atoms) { int t = ((int)(atoms * 1.08) + 63) / 64 * 64; return t < 64 ? 64 : t > EAM_ATOM_BRICK_THREADS ? EAM_ATOM_BRICK_THREADS : t; };
   // The brick: as many atoms as the workgroup has threads, the block within the 192 cells the staging covers, two workgroups per CU (80 KB of LDS each) in
   // both passes.  Tried in this order; COMD_EAM_ATOM_BRICK="by,bz" overrides.  Fixed by the first launch, the image is sized again when bricks outgrow it.
   const bool handOver = !(getenv("COMD_EAM_ATOM_HANDOVER") && atoi(getenv("COMD_EAM_ATOM_HANDOVER")) == 0) && rows <= 16 * EAM_ATOM_ROW_CHUNKS;
   if (!sim->eam_pot.atomBrickBy || !sim->eam_pot.atomBrickImageCap) {
      // (larger bricks stage fewer cells per atom and fill six waves -- and are slower: 1 x 5 x 5 1.39 ms, 1 x 4 x 6 1.42, 1 x 5 x 6 1.68 against 1.21 at 80^3; two large
      //  workgroups per CU overlap one's staging with the other's arithmetic less than three small ones)
      static const int shapes[][2] = { { 4, 4 }, { 4, 3 }, { 4, 2 }, { 2, 2 }, { 2, 1 }, { 1, 1 } };
      int ey = 0, ez = 0;
      { const char* e = getenv("COMD_EAM_ATOM_BRICK"); if (!(e && sscanf(e, "%d,%d", &ey, &ez) == 2 && ey >= 1 && ez >= 1 && 3 * (ey + 2) * (ez + 2) <= EAM_ATOM_MAX_CELLS && ey * ez <= 64)) ey = ez = 0; }
      const int nShapes = (int)(sizeof shapes / sizeof shapes[0]);
      for (int k = sim->eam_pot.atomBrickBy ? nShapes - 1 : 0; k < nShapes; ++k) {
         if (sim->eam_pot.atomBrickBy) { b.by = sim->eam_pot.atomBrickBy; b.bz = sim->eam_pot.atomBrickBz; }      // (re-sizing: the shape stays)
         else if (ey) { b.by = ey; b.bz = ez; }
         else { b.by = shapes[k][0]; b.bz = shapes[k][1]; }
         b.nby = ceilDiv(b.geom.g[1], b.by); b.nbz = ceilDiv(b.geom.g[2], b.bz);
         const int cap = eamBrickSizeImage(sim, b, st, false);
         const bool last = ey || sim->eam_pot.atomBrickBy || k == nShapes - 1;
         const int rt = rowThreadsOf(perCell * b.by * b.bz);
         const size_t lds1 = eamAtomBrickLdsBytes(1, tableDoublesOf(1), cap, rows, rt, true), lds3 = eamAtomBrickLdsBytes(3, tableDoublesOf(3), cap, rows, rt, !handOver);
         if (last || (perCell * b.by * b.bz <= 1.05 * EAM_ATOM_BRICK_THREADS && lds1 <= 80 * 1024 && lds3 <= 80 * 1024)) {
            sim->eam_pot.atomBrickBy = b.by; sim->eam_pot.atomBrickBz = b.bz; sim->eam_pot.atomBrickImageCap = cap;
            break;
         }
      }
   }
   b.by = sim->eam_pot.atomBrickBy; b.bz = sim->eam_pot.atomBrickBz;
   b.nby = ceilDiv(b.geom.g[1], b.by); b.nbz = ceilDiv(b.geom.g[2], b.bz);
   b.imageCap = sim->eam_pot.atomBrickImageCap;
   b.listRounds = rowThreadsOf(perCell * b.by * b.bz);      // (EAM_Force_atom_brick reads its row threads here)
   auto perCu = [&](size_t bytes) { return bytes > 160 * 1024 ? 0 : (int)(160 * 1024 / (((bytes + 1279) / 1280) * 1280)); };      // (the LDS is handed out in 1280-byte granules)
   if (!rowsForced) {
      const int least = ((int)(inSphere * 1.2) + 7) / 8 * 8;
      int best = rows, bestWgs = perCu(eamAtomBrickLdsBytes(1, tableDoublesOf(1), b.imageCap, rows, b.listRounds, true));
      for (int r = rows - 8; r >= least && r >= 32; r -= 8) {
         const int wgs = perCu(eamAtomBrickLdsBytes(1, tableDoublesOf(1), b.imageCap, r, b.listRounds, true));
         if (wgs > bestWgs) { bestWgs = wgs; best = r; }
      }
      rows = best;
   }
   b.rows = rows;
   if (!sim->eam_pot.brickStats) sim->eam_pot.brickStats = dalloc<int>(2);
   b.stats = sim->eam_pot.brickStats;
   b.fuseEmbed = sim->fuseEmbed; b.status = sim->status;
   { const char* e = getenv("COMD_EAM_ABLATE"); b.debug = e ? atoi(e) : 0; }
   // the rows pass 1 leaves for pass 3 (eam_atom_brick_kernels.h; COMD_EAM_ATOM_HANDOVER=0: pass 3 tests again, A/B runs)
   const int rowCap = b.listRounds <= 192 ? 256 : 512;      // atoms of a brick that can leave a row (a brick fuller than that: its last atoms walk again in pass 3)
   if (handOver && rows <= 16 * EAM_ATOM_ROW_CHUNKS) {
      const size_t nBricks = (size_t)b.geom.g[0] * b.nby * b.nbz;
      if (!sim->eam_pot.atomRows) {
         sim->eam_pot.atomRows = dalloc<unsigned>(nBricks * EAM_ATOM_ROW_CHUNKS * rowCap * 4, false);
         sim->eam_pot.atomRowCount = dalloc<unsigned short>(nBricks * rowCap * 2, false);      // (a 32-bit word per atom: the three runs' counts)
         sim->eam_pot.atomBrickSel = dalloc<unsigned long long>((size_t)sim->boxes.nLocalBoxes, false);
         HIP_CHECK(hipMemsetAsync(sim->eam_pot.atomBrickSel, 0, (size_t)sim->boxes.nLocalBoxes * sizeof(unsigned long long), st));
      }
      if (STEP == 1) sim->eam_pot.atomRowsValid = 1;
      if (STEP == 1 || sim->eam_pot.atomRowsValid) { b.rowsG = sim->eam_pot.atomRows; b.rowCountG = sim->eam_pot.atomRowCount; b.brickSel = sim->eam_pot.atomBrickSel; }
   }
   const int group = eamBrickGroupOf(sim, cells_list, num_cells, THREAD_ATOM);
   if (group) {      // the boundary / interior launch of the overlap mode: whole bricks, as cta_cell (a brick with cells of both lists would be staged twice per pass)
      eamBrickGroupLists(sim, b, st);
      b.brickList = sim->eam_pot.brickList + (group == 1 ? 0 : sim->eam_pot.brickListStride);
   } else if (cells_list) {      // a launch over any other cell list: mark the cells, every brick looks at its own (zeroed on the launch stream, see launchEamBrick)
      if (!sim->eam_pot.cellSel) {
         sim->eam_pot.cellSel = dalloc<int>((size_t)sim->boxes.nLocalBoxes, false);
         HIP_CHECK(hipMemsetAsync(sim->eam_pot.cellSel, 0, (size_t)sim->boxes.nLocalBoxes * sizeof(int), st));
      }
      b.sel = sim->eam_pot.cellSel; b.tag = ++sim->eam_pot.selTag;
      ForceTimer aux(sim, st, 1);
      hipLaunchKernelGGL(MarkCells, dim3(ceilDiv(num_cells, 256)), dim3(256), 0, st, cells_list, num_cells, sim->eam_pot.cellSel, b.tag);
   }
   // pass 3 keeps rows in the LDS only when it has none to read: what they would take is the third workgroup of a CU
   b.listQuads = ((STEP == 3 && b.rowsG) ? 0 : 1) | (rowCap << 8);
   size_t lds = eamAtomBrickLdsBytes(STEP, tableDoublesOf(STEP), b.imageCap, b.rows, b.listRounds, (b.listQuads & 1) != 0);
   { const char* e = getenv("COMD_EAM_ATOM_LDS_PAD"); if (e) lds += (size_t)atoi(e); }      // experiments: fewer workgroups per CU
   if (lds > 160 * 1024) { fprintf(stderr, "eamForce: thread_atom needs %zu bytes of LDS for this box\n", lds); exit(-1); }
   const int grid = group ? sim->eam_pot.brickCount[group - 1] : b.geom.g[0] * b.nby * b.nbz;
   if (grid <= 0) return;
   // threads: the waves that take atoms, and enough to ask for the 16 first slots of every block cell in EAM_BRICK_STAGE rounds
   const int nThreads = (b.listRounds > 256 || 3 * (b.by + 2) * (b.bz + 2) * 16 > EAM_BRICK_STAGE * 256) ? EAM_ATOM_BRICK_THREADS : 256;
   const double rcut = sim->eam_pot.cutoff * (1.0 + 4e-16);      // (as launchEamBrick: the table clamps are dead weight when every evaluated pair lies inside the tables)
   const bool clampFree = !spline && a.phi.x0 <= R(0.0) && a.rho.x0 <= R(0.0) && rcut <= (double)a.phi.xn && rcut <= (double)a.rho.xn
                          && !(getenv("COMD_EAM_CLAMP") && atoi(getenv("COMD_EAM_CLAMP")) != 0);
#define COMD_LAUNCH_EAM_ATOM_BRICK(TAB, SPL, CLP) do { \
      allowDynamicLds((const void*)EAM_Force_atom_brick<STEP, TAB, SPL, CLP>, lds); \
      hipLaunchKernelGGL((EAM_Force_atom_brick<STEP, TAB, SPL, CLP>), dim3(grid), dim3(nThreads), lds, st, a, b); } while (0)
   if (spline)           COMD_LAUNCH_EAM_ATOM_BRICK(false, true, true);
   else if (tablesInLds) { if (clampFree) COMD_LAUNCH_EAM_ATOM_BRICK(true, false, false); else COMD_LAUNCH_EAM_ATOM_BRICK(true, false, true); }
   else                  { if (clampFree) COMD_LAUNCH_EAM_ATOM_BRICK(false, false, false); else COMD_LAUNCH_EAM_ATOM_BRICK(false, false, true); }
#undef COMD_LAUNCH_EAM_ATOM_BRICK
   LAUNCH_CHECK();
}

// cta_cell: size the brick image again at the next launch (between two force evaluations only: pass 1 and pass 3 of one evaluation must stage alike)
extern "C" void comdEamBrickResize(SimGpu* sim) { sim->eam_pot.brickImageCap = 0; sim->eam_pot.atomBrickImageCap = 0; }

// what the force wrappers decided for this simulation (bench.py records it beside the numbers): {LJ thread_atom candidate lists in use (0: the 27-cell walk),
// records of the EAM brick image, Verlet-list format (comd_hip.h slabFormat), cells per EAM brick}
extern "C" void comdForcePathInfo(SimGpu* sim, int out[4])
{
   out[0] = sim->lj_pot.packedCap > 0 ? 1 : 0;
   out[1] = sim->eam_pot.brickImageCap ? sim->eam_pot.brickImageCap : sim->eam_pot.atomBrickImageCap;
   out[2] = sim->atoms.neighborList.slabFormat;
   out[3] = (sim->eam_pot.brickBy ? sim->eam_pot.brickBy * sim->eam_pot.brickBz : sim->eam_pot.atomBrickBy * sim->eam_pot.atomBrickBz) + 256 * sim->eam_pot.brickListMakes;      // (cells per brick in the low byte, times the brick lists were made above)
}

extern "C" void comdEamBrickStats(SimGpu* sim, int out[3])
{
   out[0] = out[1] = out[2] = 0;
   if (!sim->eam_pot.brickStats) return;
   int h[2];
   HIP_CHECK(hipDeviceSynchronize());
   HIP_CHECK(hipMemcpy(h, sim->eam_pot.brickStats, sizeof h, hipMemcpyDeviceToHost));
   HIP_CHECK(hipMemset(sim->eam_pot.brickStats + 1, 0, sizeof(int)));
   const bool atomBrick = !sim->eam_pot.brickBy && sim->eam_pot.atomBrickBy;      // (thread_atom on the brick image: its shape, its image)
   int by = atomBrick ? sim->eam_pot.atomBrickBy : sim->eam_pot.brickBy ? sim->eam_pot.brickBy : 4, bz = atomBrick ? sim->eam_pot.atomBrickBz : sim->eam_pot.brickBz ? sim->eam_pot.brickBz : 2;
   out[0] = h[1]; out[1] = sim->boxes.gridSize[0] * ceilDiv(sim->boxes.gridSize[1], by) * ceilDiv(sim->boxes.gridSize[2], bz);
   out[2] = atomBrick ? sim->eam_pot.atomBrickImageCap : sim->eam_pot.brickImageCap;
}

template <int STEP>
static void launchEamPair(SimGpu* sim, int num_cells, int* cells_list, int method, hipStream_t st, int spline)
{
   if (num_cells <= 0) return;
   EamArgs a = makeEamArgs(sim, num_cells, cells_list);
   ForceTimer timer(sim, st);
   if (spline) {
      // -P (gpu_kernels.cu:164-226): cubic splines in r^2 for phi and rho, coefficient tables read through L2 (16 KB each for funcfl)
      if (!a.phiS.coefficients || !a.rhoS.coefficients) { fprintf(stderr, "eamForce: spline != 0 but no spline tables were given to AllocateGpu\n"); exit(-1); }
      if ((method == THREAD_ATOM || method == WARP_ATOM) && !eamAtomBrickPath(sim, method)) {
         launchEamThreadAtom<STEP>(sim, a, num_cells, st, true);
         return;
      }
   }
   if (eamAtomBrickPath(sim, method)) {
      launchEamAtomBrick<STEP>(sim, a, num_cells, cells_list, st, spline);
      return;
   }
   if (STEP == 1) sim->eam_pot.atomRowsValid = 0;      // (another method's pass 1: the rows EAM_Force_atom_brick left are not this evaluation's)
   if (eamListedBrick(sim, method)) {
      if (sim->atoms.neighborList.nBuilds == 0) { fprintf(stderr, "the *_nl methods need buildNeighborListGpu before the first force call\n"); exit(-1); }
      launchEamBrick<STEP>(sim, a, num_cells, cells_list, st, spline, true, method);
      return;
   } else if ((method == THREAD_ATOM_NL || method == WARP_ATOM_NL) && sim->atoms.neighborList.slabFormat == 2) {
      NeighborListGpu* n = &sim->atoms.neighborList;
      (void)nlView(sim);
      NlSlabView v; v.list = n->list16; v.count = n->nNeighbors; v.rows = n->slabRows;
      const bool sameGrid = a.phi.n == a.rho.n && a.phi.x0 == a.rho.x0 && a.phi.invDx == a.rho.invDx;
      const size_t lds = eamNlLdsBytes(STEP, a.rho.n, a.phi.n, sameGrid, n->maxSlabAtoms, spline != 0);
      if (lds > 160 * 1024) { fprintf(stderr, "eamForce: %d atoms in a 27-cell stencil do not fit the LDS\n", n->maxSlabAtoms); exit(-1); }
      const int grid = ceilDiv(num_cells, EAM_NL_WAVES * 8);        // each wave walks ~8 consecutive cells
      if (spline) {
         allowDynamicLds((const void*)EAM_Force_nl_lds<STEP, true>, lds);
         hipLaunchKernelGGL((EAM_Force_nl_lds<STEP, true>), dim3(grid), dim3(64 * EAM_NL_WAVES), lds, st, a, v, n->maxSlabAtoms);
      } else {
         allowDynamicLds((const void*)EAM_Force_nl_lds<STEP, false>, lds);
         hipLaunchKernelGGL((EAM_Force_nl_lds<STEP, false>), dim3(grid), dim3(64 * EAM_NL_WAVES), lds, st, a, v, n->maxSlabAtoms);
      }
   } else if (method == THREAD_ATOM_NL || method == WARP_ATOM_NL) {
      const NlView nl = nlView(sim);
      const size_t tableBytes = eamCtaTableBytes(STEP, a.rho.n, a.phi.n);
      const unsigned nBlocks = (unsigned)ceilDiv((long)num_cells * sim->maxAtoms, 256);
      if (spline)                       hipLaunchKernelGGL((EAM_Force_thread_atom_nl<STEP, false, true>), dim3(nBlocks), dim3(256), 0, st, a, nl);
      else if (tableBytes <= 32 * 1024) hipLaunchKernelGGL((EAM_Force_thread_atom_nl<STEP, true, false>), dim3(nBlocks), dim3(256), tableBytes, st, a, nl);
      else                              hipLaunchKernelGGL((EAM_Force_thread_atom_nl<STEP, false, false>), dim3(nBlocks), dim3(256), 0, st, a, nl);
   } else if (eamBrickPath(sim, method)) {
      launchEamBrick<STEP>(sim, a, num_cells, cells_list, st, spline, false, method);
      return;
   } else if (method == CTA_CELL) {
      // COMD_EAM_CTA=cell: round 2's form, a wave stages the stencil of every cell for itself (nl_kernels.h EAM_Force_cta_cell); kept for A/B runs
      const size_t tableBytes = eamCtaTableBytes(STEP, a.rho.n, a.phi.n);
      const bool tablesInLds = !spline && tableBytes <= 32 * 1024;      // funcfl tables (500 samples) live in the LDS; setfl (10000) and spline coefficients stay in L2
      const bool sameGrid = a.phi.n == a.rho.n && a.phi.x0 == a.rho.x0 && a.phi.invDx == a.rho.invDx;
      // a stencil of 27 cells at the perfect-lattice density + 30 % (thermal crowding, cells fuller than the mean), whole staging rounds of 64
      const double cellVol = 1.0 / (sim->boxes.invBoxSize[0] * sim->boxes.invBoxSize[1] * sim->boxes.invBoxSize[2]);
      const double lat = sim->latticeConstant > 0.0 ? sim->latticeConstant : 3.615;
      const double perStencil = 27.0 * cellVol * 4.0 / (lat * lat * lat);
      int stencil = (((int)(perStencil * 1.30) + 16 + 7) / 8) * 8;
      if (stencil < 128) stencil = 128;
      if (stencil > 27 * sim->maxAtoms) stencil = ((27 * sim->maxAtoms + 7) / 8) * 8;
      if (stencil > 1024) stencil = 1024;                    // beyond that a cell takes the thread-per-atom form inside the same kernel
      // rows per atom: the cutoff sphere at that density + 50 %, a multiple of 8
      const double rc = sim->eam_pot.cutoff;
      int rows = ((int)(4.18879020478639 * rc * rc * rc * 4.0 / (lat * lat * lat) * 1.5) + 7) / 8 * 8;
      if (rows < 32) rows = 32;
      // 4 waves per workgroup, one per SIMD (5 or 6 land unevenly on the four SIMDs of a CU: measured 3.4-3.7 ms against 2.6 at 80^3)
      int waves = 4;
      { const char* e = getenv("COMD_EAM_CTA_WAVES"); if (e && atoi(e) >= 1 && atoi(e) <= 8) waves = atoi(e); }
      if (rows > 256) rows = 256;                            // the hand-over holds 16 lanes x 8 trips x 2 numbers per atom
      if (!sim->eam_pot.pairRows) {                          // first cta_cell launch: rows pass 1 leaves for pass 3
         const size_t slotsLocal = (size_t)sim->boxes.nLocalBoxes * sim->maxAtoms;
         sim->eam_pot.pairRows = dalloc<unsigned>(slotsLocal * EAM_ROW_WORDS, false);
         sim->eam_pot.pairRowCount = dalloc<unsigned short>(slotsLocal, false);
         sim->eam_pot.pairRowLen = rows;
      }
      // The LDS slice decides how many workgroups share a CU (gfx950 hands the LDS out in 1280-byte granules, 160 KB per CU), and these kernels
      // live on latency hiding: at 80^3 a slice of 384 records leaves room for two workgroups per CU in either pass, one of 376 for three
      // (measured: pass 1 1.43 -> 1.08 ms).  Shrink the slice, down to the density + 20 %, when that buys a workgroup in pass 1 or pass 3;
      // both passes must use the same size (a cell either has rows or takes the thread-per-atom form, in both).
      {
         auto perCu = [&](int step, int st) {
            const size_t b = eamCtaCellLdsBytes(step, a.rho.n, a.phi.n, tablesInLds, sameGrid, st, rows, waves);
            return b > 160 * 1024 ? 0 : (int)(160 * 1024 / (((b + 1279) / 1280) * 1280));
         };
         const int lo = (((int)(perStencil * 1.20) + 16 + 7) / 8) * 8;
         int best = stencil, bestScore = perCu(1, stencil) + perCu(3, stencil);
         for (int st = stencil - 8; st >= lo && st >= 128; st -= 8) {
            const int score = perCu(1, st) + perCu(3, st);
            if (score > bestScore) { bestScore = score; best = st; }
         }
         stencil = best;
      }
      { const char* e = getenv("COMD_EAM_STENCIL"); if (e && atoi(e) >= 64) stencil = (atoi(e) + 7) / 8 * 8; }      // experiments: LDS slice size
      const size_t lds = eamCtaCellLdsBytes(STEP, a.rho.n, a.phi.n, tablesInLds, sameGrid, stencil, rows, waves);
      if (lds > 160 * 1024) { fprintf(stderr, "eamForce: cta_cell needs %zu bytes of LDS for this box\n", lds); exit(-1); }
      const int grid = ceilDiv(num_cells, waves * 8);        // each wave walks ~8 consecutive cells
#define COMD_LAUNCH_EAM_CTA(TAB, SPL) do { \
         allowDynamicLds((const void*)EAM_Force_cta_cell<STEP, TAB, SPL>, lds); \
         hipLaunchKernelGGL((EAM_Force_cta_cell<STEP, TAB, SPL>), dim3(grid), dim3(64 * waves), lds, st, a, stencil, rows, sim->eam_pot.pairRows, sim->eam_pot.pairRowCount, sim->fuseEmbed, sim->status); } while (0)
      if (spline)           COMD_LAUNCH_EAM_CTA(false, true);
      else if (tablesInLds) COMD_LAUNCH_EAM_CTA(true, false);
      else                  COMD_LAUNCH_EAM_CTA(false, false);
#undef COMD_LAUNCH_EAM_CTA
   } else {
      launchEamThreadAtom<STEP>(sim, a, num_cells, st, false);
      return;
   }
   LAUNCH_CHECK();
}

extern "C" void eamForce1GpuAsync(SimGpu* sim, int num_cells, int* cells_list, int method, comdStream_t stream, int spline)
{ launchEamPair<1>(sim, num_cells, cells_list, method, S(stream), spline); }

extern "C" void eamForce2GpuAsync(SimGpu* sim, int num_cells, int* cells_list, int method, comdStream_t stream, int spline)
{
   (void)spline;                            /* F(rhobar) is quadratic in both modes (gpu_utility.c:443) */
   if (num_cells <= 0) return;
   if (sim->fuseEmbed && (method == CTA_CELL || eamListedBrick(sim, method) || eamAtomBrickPath(sim, method))) return;      /* eamForce1Gpu[Async] has done it for these cells (SimGpu.fuseEmbed) */
   EamArgs a = makeEamArgs(sim, num_cells, cells_list);
   // cta_cell in the overlap mode: pass 1 took whole bricks (launchEamBrick), the embedding follows the same groups over all local cells
   const int group = sim->eam_pot.brickGroup ? eamBrickGroupOf(sim, cells_list, num_cells, method) : 0;
   if (group) { a.cells = nullptr; a.nCells = num_cells = sim->boxes.nLocalBoxes; a.sel = sim->eam_pot.brickGroup; a.tag = group; }
   ForceTimer timer(sim, S(stream));
   hipLaunchKernelGGL(EAM_Force_embed, dim3(ceilDiv((long)num_cells * sim->maxAtoms, 256)), dim3(256), 0, S(stream), a);
   LAUNCH_CHECK();
}

extern "C" void eamForce3GpuAsync(SimGpu* sim, int num_cells, int* cells_list, int method, comdStream_t stream, int spline)
{ launchEamPair<3>(sim, num_cells, cells_list, method, S(stream), spline); }

extern "C" void eamForce1Gpu(SimGpu* sim, int method, int spline) { eamForce1GpuAsync(sim, sim->boxes.nLocalBoxes, nullptr, method, nullptr, spline); }
extern "C" void eamForce2Gpu(SimGpu* sim, int method, int spline) { eamForce2GpuAsync(sim, sim->boxes.nLocalBoxes, nullptr, method, nullptr, spline); }
extern "C" void eamForce3Gpu(SimGpu* sim, int method, int spline) { eamForce3GpuAsync(sim, sim->boxes.nLocalBoxes, nullptr, method, nullptr, spline); }

extern "C" void updateNeighborsGpu(SimGpu*, int*) {}
extern "C" void updateNeighborsGpuAsync(SimGpu*, int*, int, int*, comdStream_t) {}

// ---- integrator + energy -----------------------------------------------------------------------------------------
// lanes of a 256-thread workgroup that serve one cell in the integrator kernels (step_kernels.h COMD_CELL_SLOTS): the power of two that covers the usual occupancy of a
// cell of this capacity -- capacities are the lattice maximum + 10 % + 8, and cells hold about a third of that (EAM) or 60 % (LJ, multiples of 64)
static int integratorLaneBits(const SimGpu* sim)
{
   const int cap = sim->maxAtoms;
   return cap <= 48 ? 4 : cap <= 96 ? 5 : cap <= 192 ? 6 : 8;
}
static dim3 integratorGrid(const SimGpu* sim) { return dim3((unsigned)ceilDiv(sim->boxes.nLocalBoxes, 256 >> integratorLaneBits(sim))); }

extern "C" void advanceVelocityGpu(SimGpu* sim, real_t dt)
{
   const long slots = (long)sim->boxes.nLocalBoxes * sim->maxAtoms;
   hipLaunchKernelGGL(AdvanceVelocity, integratorGrid(sim), dim3(256), 0, S(sim->boundary_stream),
                      sim->atoms.p.x, sim->atoms.p.y, sim->atoms.p.z, sim->atoms.f.x, sim->atoms.f.y, sim->atoms.f.z,
                      sim->boxes.nAtoms, sim->boxes.nLocalBoxes, sim->maxAtoms, dt, integratorLaneBits(sim));
   LAUNCH_CHECK();
}

extern "C" void advancePositionGpu(SimGpu* sim, real_t dt)
{
   const long slots = (long)sim->boxes.nLocalBoxes * sim->maxAtoms;
   hipLaunchKernelGGL(AdvancePosition, integratorGrid(sim), dim3(256), 0, S(sim->boundary_stream),
                      sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z, sim->atoms.p.x, sim->atoms.p.y, sim->atoms.p.z,
                      sim->atoms.iSpecies, sim->species_mass, sim->boxes.nAtoms, sim->boxes.nLocalBoxes, sim->maxAtoms, dt, integratorLaneBits(sim));
   LAUNCH_CHECK();
}

// Verlet lists: the fused drift kernels test the displacement since the list build as they write the new positions (step_kernels.h SkinCheck);
// neighborListUpdateRequiredGpu then only reads the flag.  No lists, or none built yet: nothing to check.
static SkinCheck skinCheckOf(SimGpu* sim)
{
   NeighborListGpu* n = &sim->atoms.neighborList;
   SkinCheck sk = { nullptr, nullptr, nullptr, R(0.0), R(0.0), nullptr, nullptr, nullptr, 0, sim->status, sim->statusMirrorDev };
   sim->statusMirrored = 1;
   if (!n->lastR.x || n->nBuilds == 0 || n->forceRebuildFlag) return sk;
   sk.lastX = n->lastR.x; sk.lastY = n->lastR.y; sk.lastZ = n->lastR.z; sk.skinHalf2 = n->skinDistanceHalf2;
   sk.softHalf2 = n->softHalf2 > R(0.0) ? n->softHalf2 : n->skinDistanceHalf2;
   sk.hard = n->updateRequired + 1; sk.soft = n->updateRequired + 2; sk.progress = n->updateRequired + 3; sk.stamp = ++n->driftCount;
   n->checkFused = 1;
   return sk;
}
extern "C" void advanceVelocityPositionGpu(SimGpu* sim, real_t dtKick, real_t dtDrift)
{
   const long slots = (long)sim->boxes.nLocalBoxes * sim->maxAtoms;
   const SkinCheck sk = skinCheckOf(sim);
   hipLaunchKernelGGL(AdvanceVelocityPosition, integratorGrid(sim), dim3(256), 0, S(sim->boundary_stream),
                      sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z, sim->atoms.p.x, sim->atoms.p.y, sim->atoms.p.z,
                      sim->atoms.f.x, sim->atoms.f.y, sim->atoms.f.z, sim->atoms.iSpecies, sim->species_mass,
                      sim->boxes.nAtoms, sim->boxes.nLocalBoxes, sim->maxAtoms, dtKick, dtDrift, sk, integratorLaneBits(sim));
   LAUNCH_CHECK();
}

extern "C" void advanceVelocityVelocityPositionGpu(SimGpu* sim, real_t dtKick1, real_t dtKick2, real_t dtDrift)
{
   const long slots = (long)sim->boxes.nLocalBoxes * sim->maxAtoms;
   const SkinCheck sk = skinCheckOf(sim);
   hipLaunchKernelGGL(AdvanceVelocityVelocityPosition, integratorGrid(sim), dim3(256), 0, S(sim->boundary_stream),
                      sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z, sim->atoms.p.x, sim->atoms.p.y, sim->atoms.p.z,
                      sim->atoms.f.x, sim->atoms.f.y, sim->atoms.f.z, sim->atoms.iSpecies, sim->species_mass,
                      sim->boxes.nAtoms, sim->boxes.nLocalBoxes, sim->maxAtoms, dtKick1, dtKick2, dtDrift, sk, integratorLaneBits(sim));
   LAUNCH_CHECK();
}

extern "C" void computeEnergy(SimGpu* sim, real_t* eLocal)
{
   hipStream_t st = S(sim->boundary_stream);
   const long slots = (long)sim->boxes.nLocalBoxes * sim->maxAtoms;
   int blocks = ceilDiv(slots, 256);
   if (blocks > sim->reduceBlocks) blocks = sim->reduceBlocks;
   real_t* out = sim->reduceBuf + 2 * (size_t)sim->reduceBlocks;
   hipLaunchKernelGGL(ReduceEnergyPartial, dim3(blocks), dim3(256), 0, st, sim->atoms.e, sim->atoms.p.x, sim->atoms.p.y, sim->atoms.p.z,
                      sim->atoms.iSpecies, sim->species_mass, sim->boxes.nAtoms, sim->boxes.nLocalBoxes, sim->maxAtoms, sim->reduceBuf);
   hipLaunchKernelGGL(ReduceEnergyFinal, dim3(1), dim3(256), 0, st, sim->reduceBuf, blocks, out);
   LAUNCH_CHECK();
   HIP_CHECK(hipMemcpyAsync(sim->pinned, out, 2 * sizeof(real_t), hipMemcpyDeviceToHost, st));
   HIP_CHECK(hipStreamSynchronize(st));
   eLocal[0] = sim->pinned[0]; eLocal[1] = sim->pinned[1];
}

// ---- redistribute ---------------------------------------------------------------------------------------------------
static AtomArrays atomArrays(SimGpu* sim)
{
   AtomArrays at;
   at.rx = sim->atoms.r.x; at.ry = sim->atoms.r.y; at.rz = sim->atoms.r.z;
   at.px = sim->atoms.p.x; at.py = sim->atoms.p.y; at.pz = sim->atoms.p.z;
   at.gid = sim->atoms.gid; at.spec = sim->atoms.iSpecies;
   return at;
}

static int sortBlock(int cap) { return ((cap + 63) / 64) * 64; }

static void launchCompactSort(SimGpu* sim, int first, int nCells, hipStream_t st, bool shortRuns = false)
{
   if (nCells <= 0) return;
   // cells per wave / workgroup: the local cells are nearly all clean (one flag each), the halo cells were all just refilled
   const bool halo = shortRuns || first >= sim->boxes.nLocalBoxes;
   if (sim->maxAtoms <= 64) {
      const int run = halo ? 2 : COMPACT_RUN_WAVE;
      hipLaunchKernelGGL(CompactSortCellsWave, dim3(ceilDiv(nCells, 4 * run)), dim3(256), 0, st,
                         atomArrays(sim), sim->boxes.nAtoms, sim->cellDirty, first, nCells, sim->maxAtoms, run, sim->cellArrivals, sim->boxes.nTotalBoxes);
      LAUNCH_CHECK();
      return;
   }
   const int run = halo ? 1 : COMPACT_RUN;
   hipLaunchKernelGGL(CompactSortCells, dim3(ceilDiv(nCells, run)), dim3(sortBlock(sim->maxAtoms)), (size_t)sortBlock(sim->maxAtoms) * sizeof(int), st,
                      atomArrays(sim), sim->boxes.nAtoms, sim->cellDirty, sim->status, first, nCells, sim->maxAtoms, run, sim->cellArrivals, sim->boxes.nTotalBoxes);
   LAUNCH_CHECK();
}

extern "C" void updateLinkCellsGpu(SimGpu* sim, comdStream_t stream)
{
   hipStream_t st = S(stream);
   const int nLocal = sim->boxes.nLocalBoxes, nTotal = sim->boxes.nTotalBoxes;
   hipLaunchKernelGGL(SnapshotCells, dim3(ceilDiv(nTotal, 256)), dim3(256), 0, st, sim->boxes.nAtoms, sim->nAtomsPrev, sim->cellDirty, sim->cellArrivals, nLocal, nTotal);
   hipLaunchKernelGGL(UpdateLinkCells, dim3(ceilDiv((long)nLocal * sim->maxAtoms, 256)), dim3(256), 0, st,
                      atomArrays(sim), sim->boxes.nAtoms, sim->nAtomsPrev, sim->cellDirty, sim->status, sim->boxes, sim->maxAtoms);
   LAUNCH_CHECK();
   // local cells that lost/gained atoms and halo cells that caught migrants -- unless the host says nothing needs them compact before sortAtomsGpu (every axis of the atom
   // exchange mirrored, which skips holes, and no interior force launch in between: SimGpu.skipSortAfterUpdate)
   if (!sim->skipSortAfterUpdate) launchCompactSort(sim, 0, nTotal, st);
}

extern "C" void buildAtomListGpu(SimGpu*, comdStream_t) {}

extern "C" void sortAtomsGpu(SimGpu* sim, comdStream_t stream)
{
   // after the atom exchange: local cells that received atoms (few) and the halo cells (all of them), one launch over all cells with the short runs of the halo
   // cells (round 3: two launches; a clean local cell costs its wave one flag read)
   launchCompactSort(sim, 0, sim->boxes.nTotalBoxes, S(stream), true);
}

// ---- halo pack / unpack ------------------------------------------------------------------------------------------------
// set by comdForceScansReady(1) after scanCellListsGpu has filled every offset array the force exchange will use
extern "C" void comdForceScansReady(SimGpu* sim, int on) { sim->forceScansReady = on; }

extern "C" void getAtomMsgSoAPtr(char* buffer, AtomMsgSoA* m, int n)
{
   m->gid = (int*)(buffer + COMD_ATOM_MSG_HEADER);
   m->type = m->gid + n;
   m->rx = (real_t*)(m->type + n);
   m->ry = m->rx + n; m->rz = m->ry + n; m->px = m->rz + n; m->py = m->px + n; m->pz = m->py + n;
}

static int maxInt(int a, int b) { return a > b ? a : b; }

// both faces of an axis phase: one scan launch (two jobs), one pack launch (blockIdx.y = face)
extern "C" void compactCellsGpu2(char* workM, char* workP, const int nCells[2], int* const d_cellList[2], SimGpu* sim, int* const d_cellOffsets[2],
                                 const real_t shiftM[3], const real_t shiftP[3], const int capacityAtoms[2], comdStream_t stream)
{
   hipStream_t st = S(stream);
   const int nFaces = workP ? 2 : 1;
   ScanJobs jobs;
   AtomPackJob pj[2];
   char* work[2] = { workM, workP };
   const real_t* shift[2] = { shiftM, shiftP };
   for (int f = 0; f < 2; ++f) {
      const int k = f < nFaces ? f : 0;
      jobs.list[f] = d_cellList[k]; jobs.n[f] = nCells[k]; jobs.out[f] = d_cellOffsets[k]; jobs.total[f] = (int*)work[k];
      pj[f].msg = work[k]; pj[f].list = d_cellList[k]; pj[f].offsets = d_cellOffsets[k]; pj[f].nCells = nCells[k];
      pj[f].sx = shift[k][0]; pj[f].sy = shift[k][1]; pj[f].sz = shift[k][2]; pj[f].capacityAtoms = capacityAtoms[k];
   }
   hipLaunchKernelGGL(ScanCellCountsBatch, dim3(nFaces), dim3(1024), 0, st, sim->boxes.nAtoms, jobs);
   hipLaunchKernelGGL(LoadAtomsBufferPacked, dim3(nFaces == 2 ? maxInt(nCells[0], nCells[1]) : nCells[0], nFaces), dim3(sortBlock(sim->maxAtoms)), 0, st,
                      pj[0], pj[1], atomArrays(sim), sim->boxes.nAtoms, sim->maxAtoms, sim->status);
   LAUNCH_CHECK();
}

extern "C" void compactCellsGpu(char* work_d, int nCells, int* d_cellList, SimGpu* sim, int* d_cellOffsets,
                                const real_t shift[3], int capacityAtoms, comdStream_t stream)
{
   const int n[2] = { nCells, 0 }; int* const lists[2] = { d_cellList, nullptr }; int* const offs[2] = { d_cellOffsets, nullptr };
   const int caps[2] = { capacityAtoms, 0 };
   compactCellsGpu2(work_d, nullptr, n, lists, sim, offs, shift, shift, caps, stream);
}

extern "C" void comdReadDeviceInt2(const int* d_a, const int* d_b, int out[2], comdStream_t stream)
{
   HIP_CHECK(hipMemcpyAsync(&out[0], d_a, sizeof(int), hipMemcpyDeviceToHost, S(stream)));
   HIP_CHECK(hipMemcpyAsync(&out[1], d_b, sizeof(int), hipMemcpyDeviceToHost, S(stream)));
   HIP_CHECK(hipStreamSynchronize(S(stream)));
}

extern "C" void comdMirrorCounts(const int* d0, const int* d1, const int* d2, const int* d3, int* pinnedDst, comdStream_t stream)
{
   hipLaunchKernelGGL(MirrorCounts, dim3(1), dim3(64), 0, S(stream), d0, d1, d2, d3, pinnedDst);
   LAUNCH_CHECK();
}

extern "C" int atomMsgCountGpu(SimGpu* sim, const char* msg_d, comdStream_t stream)
{
   (void)sim;
   return comdReadDeviceInt((const int*)msg_d, stream);
}

// msgB == NULL: one message
extern "C" void unloadAtomsBufferToGpu2(const char* msgA, int nBufA, int maxAtomsA, const char* msgB, int nBufB, int maxAtomsB, SimGpu* sim, comdStream_t stream)
{
   const int boundA = nBufA >= 0 ? nBufA : maxAtomsA, boundB = msgB ? (nBufB >= 0 ? nBufB : maxAtomsB) : 0;
   const int bound = maxInt(boundA, boundB);
   if (bound <= 0) return;
   AtomUnpackJob a = { msgA, nBufA, maxAtomsA }, b = { msgB ? msgB : msgA, msgB ? nBufB : 0, msgB ? maxAtomsB : 0 };
   hipLaunchKernelGGL(UnloadAtomsBufferPacked, dim3(ceilDiv(bound, 256), msgB ? 2 : 1), dim3(256), 0, S(stream), a, b,
                      atomArrays(sim), sim->boxes.nAtoms, sim->cellDirty, sim->status, sim->boxes, sim->maxAtoms);
   LAUNCH_CHECK();
}

extern "C" void unloadAtomsBufferToGpu(const char* msg_d, int nBuf, int maxAtomsInMsg, SimGpu* sim, comdStream_t stream)
{
   unloadAtomsBufferToGpu2(msg_d, nBuf, maxAtomsInMsg, nullptr, 0, 0, sim, stream);
}

extern "C" void scanCellListsGpu(SimGpu* sim, int nLists, int** d_cellLists, const int* nCells, int** d_cellOffsets, comdStream_t stream)
{
   if (nLists < 1 || nLists > 12) { fprintf(stderr, "scanCellListsGpu: 1..12 lists per call\n"); exit(-1); }
   ScanJobs jobs;
   for (int i = 0; i < 12; ++i) { const int k = i < nLists ? i : 0; jobs.list[i] = d_cellLists[k]; jobs.n[i] = nCells[k]; jobs.out[i] = d_cellOffsets[k]; jobs.total[i] = nullptr; }
   hipLaunchKernelGGL(ScanCellCountsBatch, dim3(nLists), dim3(1024), 0, S(stream), sim->boxes.nAtoms, jobs);
   LAUNCH_CHECK();
}

static SlotJob slotJob(const real_t* buf, int nCells, int* list, int* offsets, int bound, const real_t* shift)
{
   SlotJob j; j.buf = (real_t*)buf; j.list = list; j.offsets = offsets; j.nCells = nCells; j.boundAtoms = bound;
   j.sx = shift ? shift[0] : 0.0; j.sy = shift ? shift[1] : 0.0; j.sz = shift ? shift[2] : 0.0;
   return j;
}

// scan the listed cells' occupancies unless comdForceScansReady(1) says the batched scan of the step already did
static void scanIfNeeded(SimGpu* sim, int nFaces, const int* nCells, int* const* lists, int* const* offsets, hipStream_t st)
{
   if (sim->forceScansReady) return;
   ScanJobs jobs;
   for (int i = 0; i < 12; ++i) { const int k = i < nFaces ? i : 0; jobs.list[i] = lists[k]; jobs.n[i] = nCells[k]; jobs.out[i] = offsets[k]; jobs.total[i] = nullptr; }
   hipLaunchKernelGGL(ScanCellCountsBatch, dim3(nFaces), dim3(1024), 0, st, sim->boxes.nAtoms, jobs);
}

// kind 0: dF/drho (1 real per atom), 1: positions + shift (3 reals per atom); bufP == NULL: one face
static void loadSlotBuffers(int kind, real_t* bufM, real_t* bufP, const int nCells[2], int* const lists[2], int* const offsets[2], const int bounds[2],
                            const real_t* shiftM, const real_t* shiftP, SimGpu* sim, hipStream_t st)
{
   const int nFaces = bufP ? 2 : 1;
   scanIfNeeded(sim, nFaces, nCells, lists, offsets, st);
   const SlotJob a = slotJob(bufM, nCells[0], lists[0], offsets[0], bounds[0], shiftM);
   const SlotJob b = bufP ? slotJob(bufP, nCells[1], lists[1], offsets[1], bounds[1], shiftP) : a;
   const dim3 grid(nFaces == 2 ? maxInt(nCells[0], nCells[1]) : nCells[0], nFaces), block(sortBlock(sim->maxAtoms));
   if (kind == 0) hipLaunchKernelGGL(LoadForceBuffer, grid, block, 0, st, a, b, sim->eam_pot.dfEmbed, sim->boxes.nAtoms, sim->maxAtoms, sim->status);
   else           hipLaunchKernelGGL(LoadPositionBuffer, grid, block, 0, st, a, b, sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z, sim->boxes.nAtoms, sim->maxAtoms, sim->status);
   LAUNCH_CHECK();
}

static void unloadSlotBuffers(int kind, const real_t* bufA, const real_t* bufB, const int nCells[2], int* const lists[2], int* const offsets[2],
                              SimGpu* sim, hipStream_t st)
{
   const int nFaces = bufB ? 2 : 1;
   scanIfNeeded(sim, nFaces, nCells, lists, offsets, st);
   const SlotJob a = slotJob(bufA, nCells[0], lists[0], offsets[0], 0, nullptr);
   const SlotJob b = bufB ? slotJob(bufB, nCells[1], lists[1], offsets[1], 0, nullptr) : a;
   const dim3 grid(nFaces == 2 ? maxInt(nCells[0], nCells[1]) : nCells[0], nFaces), block(sortBlock(sim->maxAtoms));
   if (kind == 0) hipLaunchKernelGGL(UnloadForceBuffer, grid, block, 0, st, a, b, sim->eam_pot.dfEmbed, sim->boxes.nAtoms, sim->maxAtoms);
   else           hipLaunchKernelGGL(UnloadPositionBuffer, grid, block, 0, st, a, b, sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z, sim->boxes.nAtoms, sim->maxAtoms);
   LAUNCH_CHECK();
}

extern "C" void loadForceBufferFromGpu(real_t* gpu_buf, int nCells, int* d_cellList, int* d_cellOffsets, SimGpu* sim, comdStream_t stream)
{
   const int n[2] = { nCells, 0 }; int* const l[2] = { d_cellList, nullptr }; int* const o[2] = { d_cellOffsets, nullptr }; const int b[2] = { sim->msgBoundAtoms, 0 };
   loadSlotBuffers(0, gpu_buf, nullptr, n, l, o, b, nullptr, nullptr, sim, S(stream));
}

extern "C" void unloadForceBufferToGpu(const real_t* gpu_buf, int nCells, int* d_cellList, int* d_cellOffsets, SimGpu* sim, comdStream_t stream)
{
   const int n[2] = { nCells, 0 }; int* const l[2] = { d_cellList, nullptr }; int* const o[2] = { d_cellOffsets, nullptr };
   unloadSlotBuffers(0, gpu_buf, nullptr, n, l, o, sim, S(stream));
}

extern "C" void loadForceBufferFromGpu2(real_t* bufM, real_t* bufP, const int nCells[2], int* const d_cellList[2], int* const d_cellOffsets[2],
                                        const int boundAtoms[2], SimGpu* sim, comdStream_t stream)
{ loadSlotBuffers(0, bufM, bufP, nCells, d_cellList, d_cellOffsets, boundAtoms, nullptr, nullptr, sim, S(stream)); }

extern "C" void unloadForceBufferToGpu2(const real_t* bufA, const real_t* bufB, const int nCells[2], int* const d_cellList[2], int* const d_cellOffsets[2],
                                        SimGpu* sim, comdStream_t stream)
{ unloadSlotBuffers(0, bufA, bufB, nCells, d_cellList, d_cellOffsets, sim, S(stream)); }

extern "C" void mirrorAtomCellsGpu(const int nCells[2], int* const d_cellList[2], const real_t shiftM[3], const real_t shiftP[3], int firstAxis, int axis,
                                   SimGpu* sim, comdStream_t stream)
{
   MirrorAtomsJob jb;
   for (int f = 0; f < 2; ++f) {
      const real_t* sh = f ? shiftP : shiftM;
      jb.list[f] = d_cellList[f]; jb.nCells[f] = nCells[f]; jb.sx[f] = sh[0]; jb.sy[f] = sh[1]; jb.sz[f] = sh[2];
   }
   const int most = maxInt(nCells[0], nCells[1]);
   if (most <= 0) return;
   hipLaunchKernelGGL(MirrorAtomCells, dim3(ceilDiv((long)most * sim->maxAtoms, 256), 2), dim3(256), 0, S(stream), jb, atomArrays(sim), sim->boxes.nAtoms,
                      sim->cellArrivals, sim->boxes.nTotalBoxes, firstAxis, axis, sim->cellDirty, sim->status, sim->boxes, sim->maxAtoms);
   LAUNCH_CHECK();
}

extern "C" void mirrorSlotCellsGpu(int kind, int nPairs, const int* d_dst, const int* d_src, const real_t* d_shift, SimGpu* sim, comdStream_t stream)
{
   if (nPairs <= 0) return;
   hipLaunchKernelGGL(MirrorSlotCells, dim3(nPairs), dim3(sortBlock(sim->maxAtoms)), 0, S(stream), kind, nPairs, d_dst, d_src, d_shift,
                      sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z, sim->eam_pot.dfEmbed, sim->boxes.nAtoms, sim->maxAtoms);
   LAUNCH_CHECK();
}

// ---- Verlet neighbour lists ------------------------------------------------------------------------------------------------
extern "C" void emptyNeighborListGpu(SimGpu* sim, int)
{
   NeighborListGpu* n = &sim->atoms.neighborList;
   if (n->nNeighbors) HIP_CHECK(hipMemsetAsync(n->nNeighbors, 0, (size_t)sim->boxes.nLocalBoxes * sim->maxAtoms * (n->slabFormat == 1 ? NL_GROUPS : 1) * sizeof(int), S(sim->boundary_stream)));
}

extern "C" void neighborListForceRebuildGpu(NeighborListGpu* nl) { nl->forceRebuildFlag = 1; }      // gpu_neighborList.c:88-93, same signature

extern "C" int neighborListUpdateRequiredGpu(SimGpu* sim)
{
   NeighborListGpu* n = &sim->atoms.neighborList;
   if ((!n->list && !n->list16 && !n->pairlist && !n->brickRows) || n->forceRebuildFlag) return 1;
   hipStream_t st = S(sim->boundary_stream);
   // The flags live in pinned host memory (the kernels write them there): a stream synchronisation and a host read, no copy.
   if (n->checkFused) {                                      // the drift kernel of the step checked the positions it wrote (step_kernels.h SkinCheck)
      n->checkFused = 0;
      HIP_CHECK(hipStreamSynchronize(st));
      return ((volatile int*)n->updateRequiredHost)[1] > n->buildDrift ? 1 : 0;
   }
   HIP_CHECK(hipStreamSynchronize(st));
   *n->updateRequiredHost = 0;                               // (raised only; cleared here, when nothing that writes it is in flight)
   hipLaunchKernelGGL(NeighborListUpdateRequired, dim3(ceilDiv((long)sim->boxes.nLocalBoxes * sim->maxAtoms, 256)), dim3(256), 0, st,
                      sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z, n->lastR.x, n->lastR.y, n->lastR.z,
                      sim->boxes.nAtoms, sim->boxes.nLocalBoxes, sim->maxAtoms, n->skinDistanceHalf2, n->updateRequired);
   LAUNCH_CHECK();
   HIP_CHECK(hipStreamSynchronize(st));
   const int v = *(volatile int*)n->updateRequiredHost;
   *n->updateRequiredHost = 0;
   return v;
}

extern "C" int comdNeighborListUpdateDeferredGpu(SimGpu* sim)
{
   NeighborListGpu* n = &sim->atoms.neighborList;
   const char* envSync = getenv("COMD_NL_SYNC");
   const bool forceSync = envSync && atoi(envSync) != 0;
   if (forceSync || !n->checkFused || n->forceRebuildFlag || n->lastInterval <= 0 || n->softHalf2 <= R(0.0)) return neighborListUpdateRequiredGpu(sim);
   n->checkFused = 0;
   const int G = n->driftCount, k = G - n->buildDrift;       // drift kernels so far, since the build
   const volatile int* f = (volatile int*)n->updateRequiredHost;
   // what drifts up to G - 2 found must be known: drift kernel G - 1 says so as it starts (f[3] = G - 2).  It has started long ago unless the host ran ahead of the
   // device by more than a step; then wait -- on the pinned word, not on the stream
   if (k >= 3) {
      long spins = 0;
      while (f[3] < G - 2) { __builtin_ia32_pause(); if (++spins > 200000000L) { HIP_CHECK(hipStreamSynchronize(S(sim->boundary_stream))); break; } }
   }
   const int hard = f[1], soft = f[2];
   if (hard > n->buildDrift && hard < G) {                   // the force evaluation behind drift `hard` used these lists beyond skin/2
      fprintf(stderr, "Rank %d: an atom moved more than skin/2 between two displacement tests of the deferred list check (margin %.1f %% of skin/2 after lists that lasted %d steps): "
                      "the neighbour lists were used beyond their validity.  COMD_NL_SYNC=1 tests every step before the force (blocking).\n",
              g_rank, 100.0 * (1.0 - sqrt((double)n->softHalf2 / (double)n->skinDistanceHalf2)), n->lastInterval);
      exit(-1);
   }
   return soft > n->buildDrift ? 1 : 0;
}

static void buildNeighborListImpl(SimGpu* sim, int method, int boundaryFlag)
{
   (void)method; (void)boundaryFlag;
   NeighborListGpu* n = &sim->atoms.neighborList;
   if (n->slabFormat == 3) {                   // pairlists: remember where the atoms are; the next force call generates the bits
      const size_t bytes = (size_t)sim->boxes.nLocalBoxes * sim->maxAtoms * sizeof(real_t);
      hipStream_t st = S(sim->boundary_stream);
      HIP_CHECK(hipMemcpyAsync(n->lastR.x, sim->atoms.r.x, bytes, hipMemcpyDeviceToDevice, st));
      HIP_CHECK(hipMemcpyAsync(n->lastR.y, sim->atoms.r.y, bytes, hipMemcpyDeviceToDevice, st));
      HIP_CHECK(hipMemcpyAsync(n->lastR.z, sim->atoms.r.z, bytes, hipMemcpyDeviceToDevice, st));
      n->forceRebuildFlag = 0;
      n->nBuilds++;
      return;
   }
   if (!n->list && !n->list16 && !n->brickRows) { fprintf(stderr, "buildNeighborListGpu: no lists allocated (GpuConfig.skinDistance == 0)\n"); exit(-1); }
   const real_t cutoff = sim->do_eam ? sim->eam_pot.cutoff : sim->lj_pot.cutoff;
   const real_t rBuild = cutoff + n->skinDistance;
   if (n->slabFormat == 4) {
      // rows of the brick kernel: the cells were just re-binned, so the image is sized again for the fullest block; STEP 0 sweeps every brick once
      hipStream_t st = S(sim->boundary_stream);
      // The brick lists (and the image size they were made for) are kept from build to build: making them costs the host a read-back of the occupancies and 1 ms in
      // which the device idles.  What tells when they are stale is the build kernel itself -- it counts the bricks whose block no longer fits the image (they take the
      // thread-per-atom form, correct and slow) -- read back asynchronously and looked at by the NEXT build, which then makes the lists again.
      if (!n->brickStatsMirror) { HIP_CHECK(hipHostMalloc((void**)&n->brickStatsMirror, 64, hipHostMallocDefault)); memset(n->brickStatsMirror, 0, 64); }
      if (n->brickStatsEvent) {
         HIP_CHECK(hipEventSynchronize((hipEvent_t)n->brickStatsEvent));      // recorded a whole list life ago
         if (n->brickStatsMirror[1] > 0) sim->eam_pot.brickListsValid = 0;
      }
      if (getenv("COMD_EAM_LISTS_EVERY_BUILD")) sim->eam_pot.brickListsValid = 0;      // (A/B runs, tests)
      if (!sim->eam_pot.brickListsValid) {
         eamBrickBuildLists(sim, st, sim->eam_pot.phiS.coefficients != nullptr);      // (sizes the image as well)
         sim->eam_pot.brickListsValid = 1;
      }
      EamArgs a = makeEamArgs(sim, sim->boxes.nLocalBoxes, nullptr);
      if (!sim->eam_pot.brickStats) sim->eam_pot.brickStats = dalloc<int>(2);
      HIP_CHECK(hipMemsetAsync(sim->eam_pot.brickStats, 0, 2 * sizeof(int), st));
      launchEamBrick<0>(sim, a, sim->boxes.nLocalBoxes, nullptr, st, 0, true, THREAD_ATOM_NL);
      HIP_CHECK(hipMemcpyAsync(n->brickStatsMirror, sim->eam_pot.brickStats, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
      if (!n->brickStatsEvent) { hipEvent_t e; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); n->brickStatsEvent = (void*)e; }
      HIP_CHECK(hipEventRecord((hipEvent_t)n->brickStatsEvent, st));
      const size_t bytes = (size_t)sim->boxes.nLocalBoxes * sim->maxAtoms * sizeof(real_t);
      HIP_CHECK(hipMemcpyAsync(n->lastR.x, sim->atoms.r.x, bytes, hipMemcpyDeviceToDevice, st));
      HIP_CHECK(hipMemcpyAsync(n->lastR.y, sim->atoms.r.y, bytes, hipMemcpyDeviceToDevice, st));
      HIP_CHECK(hipMemcpyAsync(n->lastR.z, sim->atoms.r.z, bytes, hipMemcpyDeviceToDevice, st));
      n->forceRebuildFlag = 0;
      n->nBuilds++;
      return;
   }
   if (n->slabFormat) {
      hipStream_t st = S(sim->boundary_stream);
      NlSlabView sv; sv.list = n->list16; sv.count = n->nNeighbors; sv.rows = n->slabRows;
      HIP_CHECK(hipMemsetAsync(n->stats, 0, 2 * sizeof(int), st));
      if (n->slabFormat == 2) {
         // A wave stages a whole 27-cell stencil; the LDS it needs decides how many waves share a CU.  Worst case is 27 full cells
         // (1188 atoms at 44 slots: one workgroup per CU); what the last build saw (+25 %) is usually 2-3x less.  If a stencil does not
         // fit, the kernel says so through stats[0] and the build is repeated with the worst-case size.
         const int worst = 27 * sim->maxAtoms < 1536 ? 27 * sim->maxAtoms : 1536;
         int stencilCap = n->nBuilds > 0 && n->maxSlabAtoms > 0 ? n->maxSlabAtoms + n->maxSlabAtoms / 4 + 16 : worst;
         if (stencilCap > worst) stencilCap = worst;
         for (;;) {
            const size_t lds = (size_t)EAM_NL_WAVES * eamBuildWaveBytes(stencilCap, n->slabRows);
            allowDynamicLds((const void*)BuildNeighborListCell16, lds);
            HIP_CHECK(hipMemsetAsync(n->stats, 0, 2 * sizeof(int), st));
            hipLaunchKernelGGL(BuildNeighborListCell16, dim3(ceilDiv(sim->boxes.nLocalBoxes, EAM_NL_WAVES * 8)), dim3(64 * EAM_NL_WAVES), lds, st,
                               sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z, sim->boxes.nAtoms, sim->neighbor_cells,
                               sim->boxes.nLocalBoxes, sim->maxAtoms, sv, rBuild * rBuild, n->lastR.x, n->lastR.y, n->lastR.z, n->stats, sim->status, stencilCap);
            LAUNCH_CHECK();
            int h[2];
            HIP_CHECK(hipMemcpyAsync(h, n->stats, sizeof h, hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
            n->maxSlabAtoms = h[0]; n->maxCellAtoms = h[1] > 0 ? h[1] : 1;
            if (h[0] <= stencilCap || stencilCap == worst) break;
            stencilCap = worst;
         }
         n->forceRebuildFlag = 0;
         n->nBuilds++;
         return;
      } else
      {
         // one workgroup per cell, a thread per slot; the LDS holds a whole group of full cells (<= 9 * 512 atoms = 108 KB)
         // COMD_NL_BANK_ORDER=1: rows ordered by LDS bank class (nl_kernels.h): 2 x 16 counters of 16 bits per thread behind the records.  An experiment that settled a
         // question (profiles/r04_experiments/README.md): it takes 45 % of the bank-conflict cycles out of LJ_Force_nl_slabs and 1.3 % of its time -- the kernel is bound by
         // VALU issue, not by the LDS -- while the build goes from 11 to 79 ms.  Off by default.
         const int bankOrder = getenv("COMD_NL_BANK_ORDER") && atoi(getenv("COMD_NL_BANK_ORDER")) != 0;
         const int groupCap = NL_GROUP_CELLS * sim->maxAtoms;
         const size_t lds = (size_t)3 * groupCap * sizeof(real_t) + (bankOrder ? (size_t)2 * 16 * sim->maxAtoms * sizeof(unsigned short) : 0);
         allowDynamicLds((const void*)BuildNeighborListSlabs, lds);
         hipLaunchKernelGGL(BuildNeighborListSlabs, dim3(sim->boxes.nLocalBoxes), dim3(sim->maxAtoms), lds, st,
                            sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z, sim->boxes.nAtoms, sim->neighbor_cells,
                            sim->boxes.nLocalBoxes, sim->maxAtoms, sv, rBuild * rBuild, n->lastR.x, n->lastR.y, n->lastR.z, n->stats, sim->status, bankOrder, groupCap);
      }
      LAUNCH_CHECK();
      int h[2];
      HIP_CHECK(hipMemcpyAsync(h, n->stats, sizeof h, hipMemcpyDeviceToHost, st));      // builds are rare: one blocking read each
      HIP_CHECK(hipStreamSynchronize(st));
      n->maxSlabAtoms = h[0]; n->maxCellAtoms = h[1] > 0 ? h[1] : 1;
      if (n->slabFormat == 1 && n->maxCellAtoms > 512) { fprintf(stderr, "buildNeighborListGpu: %d atoms in a cell, the slab kernel takes 512\n", n->maxCellAtoms); exit(-1); }
      n->forceRebuildFlag = 0;
      n->nBuilds++;
      return;
   }
   NlView v; v.list = n->list; v.count = n->nNeighbors; v.maxNbr = n->maxNeighbors;
   hipLaunchKernelGGL(BuildNeighborList, dim3(ceilDiv((long)sim->boxes.nLocalBoxes * sim->maxAtoms, 256)), dim3(256), 0, S(sim->boundary_stream),
                      sim->atoms.r.x, sim->atoms.r.y, sim->atoms.r.z, sim->boxes.nAtoms, sim->neighbor_cells, (const int*)nullptr,
                      sim->boxes.nLocalBoxes, sim->maxAtoms, v, rBuild * rBuild, n->lastR.x, n->lastR.y, n->lastR.z, sim->status);
   LAUNCH_CHECK();
   n->forceRebuildFlag = 0;
   n->nBuilds++;
}

extern "C" void buildNeighborListGpu(SimGpu* sim, int method, int boundaryFlag)
{
   NeighborListGpu* n = &sim->atoms.neighborList;
   const int had = n->nBuilds;
   buildNeighborListImpl(sim, method, boundaryFlag);
   // bookkeeping of the deferred displacement test (comdNeighborListUpdateDeferredGpu): how long the previous lists lasted sizes the margin of these
   n->lastInterval = had > 0 ? n->driftCount - n->buildDrift : 0;
   n->buildDrift = n->driftCount;
   double frac = n->lastInterval > 0 ? 4.0 / n->lastInterval : 0.0;      // two steps' worth of twice the average growth per step
   if (frac < 0.1) frac = 0.1;
   if (frac > 0.6) frac = 0.6;
   { const char* e = getenv("COMD_NL_MARGIN"); if (e && atof(e) >= 0.0 && atof(e) < 1.0) frac = atof(e); }      // tests: force the stop
   n->softHalf2 = n->lastInterval > 0 ? (real_t)((1.0 - frac) * (1.0 - frac)) * n->skinDistanceHalf2 : R(0.0);
}

extern "C" int pairlistUpdateRequiredGpu(SimGpu* sim) { return neighborListUpdateRequiredGpu(sim); }
extern "C" void comdPairlistGenerated(SimGpu* sim) { sim->atoms.neighborList.pairlistBuildId = sim->atoms.neighborList.nBuilds; }

// the reference keeps a gid -> slot hash table for its list mode (hashTable.c); here atoms keep their slots between builds
extern "C" void initHashTableGpu(HashTableGpu* hashTable, int nMaxEntries) { if (hashTable) { hashTable->nMaxEntries = nMaxEntries; hashTable->nEntriesPut = hashTable->nEntriesGet = 0; } }
extern "C" void emptyHashTableGpu(HashTableGpu* hashTable) { if (hashTable) hashTable->nEntriesPut = hashTable->nEntriesGet = 0; }

// comm.h:40-74 (libmp / GPUDirect-Async): not part of this library -- "not in use" answers so that the reference's host objects take
// their plain send/receive path (haloExchange.c:726-730), which include/comd_hip.h serves through CommTransport
extern "C" int  comm_use_comm(void) { return 0; }
extern "C" int  comm_use_gdrdma(void) { return 0; }
extern "C" int  comm_use_async(void) { return 0; }
extern "C" int  comm_use_gpu_comm(void) { return 0; }
extern "C" int  comm_select_device(int mpiRank) { (void)mpiRank; return 0; }
extern "C" int  comm_init(...) { return 0; }
extern "C" void comm_finalize(void) {}

// The comm-layer-only part of the reference's link surface: defined so that its host objects link, never reachable while
// comm_use_comm() / comm_use_async() answer 0.  A call is a bug in the caller: say which symbol and stop.
#define COMD_ABSENT(name) extern "C" int name(...) { fprintf(stderr, "%s: the libmp / GPUDirect-Async layer is not part of this build (comm_use_comm() is 0)\n", #name); exit(-1); }
COMD_ABSENT(comm_irecv) COMD_ABSENT(comm_isend) COMD_ABSENT(comm_isend_on_stream) COMD_ABSENT(comm_send_ready) COMD_ABSENT(comm_send_ready_on_stream)
COMD_ABSENT(comm_wait_ready_on_stream) COMD_ABSENT(comm_wait_all) COMD_ABSENT(comm_wait_all_on_stream) COMD_ABSENT(comm_flush) COMD_ABSENT(comm_progress)
COMD_ABSENT(loadAtomsBufferFromGpu_Async) COMD_ABSENT(loadAtomsBufferFromGpu_Comm) COMD_ABSENT(unloadAtomsBufferToGpu_Async) COMD_ABSENT(unloadAtomsBufferToGpu_Comm)
COMD_ABSENT(loadForceBufferFromGpu_Async) COMD_ABSENT(loadForceBufferFromGpu_Comm) COMD_ABSENT(unloadForceBufferToGpu_Async) COMD_ABSENT(unloadForceBufferToGpu_Comm)
COMD_ABSENT(unloadForceScanCells) COMD_ABSENT(exchangeDataForceGpu_KI) COMD_ABSENT(neighborListUpdateRequiredGpu_Async)
#undef COMD_ABSENT

extern "C" void loadPositionBufferFromGpu(real_t* gpu_buf, int nCells, int* d_cellList, int* d_cellOffsets, const real_t shift[3], SimGpu* sim, comdStream_t stream)
{
   const int n[2] = { nCells, 0 }; int* const l[2] = { d_cellList, nullptr }; int* const o[2] = { d_cellOffsets, nullptr }; const int b[2] = { sim->msgBoundAtoms, 0 };
   const int ready = sim->forceScansReady; sim->forceScansReady = 1;        // the offsets of a position exchange are those of the last list build
   loadSlotBuffers(1, gpu_buf, nullptr, n, l, o, b, shift, nullptr, sim, S(stream));
   sim->forceScansReady = ready;
}

extern "C" void unloadPositionBufferToGpu(const real_t* gpu_buf, int nCells, int* d_cellList, int* d_cellOffsets, SimGpu* sim, comdStream_t stream)
{
   const int n[2] = { nCells, 0 }; int* const l[2] = { d_cellList, nullptr }; int* const o[2] = { d_cellOffsets, nullptr };
   const int ready = sim->forceScansReady; sim->forceScansReady = 1;
   unloadSlotBuffers(1, gpu_buf, nullptr, n, l, o, sim, S(stream));
   sim->forceScansReady = ready;
}

extern "C" void loadPositionBufferFromGpu2(real_t* bufM, real_t* bufP, const int nCells[2], int* const d_cellList[2], int* const d_cellOffsets[2],
                                           const int boundAtoms[2], const real_t shiftM[3], const real_t shiftP[3], SimGpu* sim, comdStream_t stream)
{
   const int ready = sim->forceScansReady; sim->forceScansReady = 1;
   loadSlotBuffers(1, bufM, bufP, nCells, d_cellList, d_cellOffsets, boundAtoms, shiftM, shiftP, sim, S(stream));
   sim->forceScansReady = ready;
}

extern "C" void unloadPositionBufferToGpu2(const real_t* bufA, const real_t* bufB, const int nCells[2], int* const d_cellList[2], int* const d_cellOffsets[2],
                                           SimGpu* sim, comdStream_t stream)
{
   const int ready = sim->forceScansReady; sim->forceScansReady = 1;
   unloadSlotBuffers(1, bufA, bufB, nCells, d_cellList, d_cellOffsets, sim, S(stream));
   sim->forceScansReady = ready;
}
