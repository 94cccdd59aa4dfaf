/* reference_callsites.c -- TEST INFRASTRUCTURE (tests/test_boundary_shim.py compiles and links it, and runs it on the GPU through shim_driver.c).
 *
 * The drop-in claim of include/comd_hip_shim.h, put through a compiler: every device-library call expression of the reference's
 * hot-path host files, written here AS THE REFERENCE WRITES IT (argument for argument; the file:line each one comes from is in the
 * comment beside it), inside scaffolding of our own.  `SimFlat` below mirrors only the fields of CoMDTypes.h:75-135 that those
 * expressions touch; with the reference's own headers the same expressions see the same names.
 */
#include <stddef.h>
#include "comd_hip.h"

/* the fields of the reference's SimFlat / LinkCell / BasePotential / exchange parameter blocks that the call sites name */
typedef struct { int nLocalBoxes, nTotalBoxes; } LinkCell;
typedef struct { real_t cutoff; } BasePotential;
typedef struct SimFlatSt {
   LinkCell* boxes;
   BasePotential* pot;
   SimGpu gpu;
   int method, ljInterpolation, spline, usePairlist, gpuAsync, gpuProfile;
   real_t skinDistance;
   int n_boundary_cells;
   int *boundary_cells, *interior_cells;
   int* flags;
   char *gpu_atoms_buf, *gpu_force_buf;
   comdStream_t boundary_stream, interior_stream;      /* cudaStream_t in the reference (CoMDTypes.h:116-117) */
   /* scaffolding: where the reference calls haloExchange(sim->atomExchange, sim) (timestep.c:269) and haloExchange(pot->forceExchange, s) (eam.c:241) --
    * host code of the reference, not a device-library call -- the driver hooks its own exchange loop built from the four pack/unpack scaffolds below */
   void (*atomHalo)(struct SimFlatSt*);
   void (*forceHalo)(struct SimFlatSt*);
   void* world;
} SimFlat;
typedef struct { int nCells[6]; int* cellListGpu[6]; int *d_natoms_buf, *d_partial_sums; } AtomExchangeParms;
typedef struct { int nCells[6]; int *sendCellsGpu[6], *recvCellsGpu[6], *natoms_buf[6], *partial_sums[6]; } ForceExchangeParms;
typedef struct { int gid, type; real_t rx, ry, rz, px, py, pz; } AtomMsg;       /* haloExchange.h:32-38 */
typedef struct { real_t dfEmbed; } ForceMsg;

#define COMD_SHIM_SIM sim
#include "comd_hip_shim.h"

int shim_ljForce(SimFlat* sim)
{
   ljForceGpu(&(sim->gpu), sim->ljInterpolation, sim->gpu.boxes.nLocalBoxes, NULL, sim->pot->cutoff + sim->skinDistance, sim->method);   /* ljForce.c:141 */
   return 0;
}

int shim_eamForceGpu(SimFlat* s)
{
   if (s->gpuAsync) {
      updateNeighborsGpuAsync(s->gpu, s->flags, s->n_boundary_cells, s->boundary_cells, s->boundary_stream);                               /* eam.c:204 */
      eamForce1GpuAsync(s->gpu, s->gpu.b_list, s->n_boundary_cells, s->boundary_cells, s->method, s->boundary_stream, s->spline);          /* eam.c:207 */
      eamForce2GpuAsync(s->gpu, s->gpu.b_list, s->n_boundary_cells, s->boundary_cells, s->method, s->boundary_stream, s->spline);          /* eam.c:208 */
      cudaStreamSynchronize(s->boundary_stream);                                                                                           /* eam.c:211 */
      int n_interior_cells = s->gpu.boxes.nLocalBoxes - s->n_boundary_cells;
      eamForce3GpuAsync(s->gpu, s->gpu.i_list, n_interior_cells, s->interior_cells, s->method, s->interior_stream, s->spline);             /* eam.c:215 */
   } else {
      updateNeighborsGpu(s->gpu, s->flags);                                                                                                /* eam.c:220 */
      eamForce1Gpu(s->gpu,s->method, s->spline);                                                                                           /* eam.c:232 */
      if (!s->gpuProfile)
         eamForce2Gpu(s->gpu,s->method, s->spline);                                                                                        /* eam.c:235 */
   }
   if (!s->gpuProfile) {
      if (s->forceHalo) s->forceHalo(s);                                                                                                   /* eam.c:241 haloExchange(pot->forceExchange, s) */
      if (s->gpuAsync) {
         cudaStreamSynchronize(s->boundary_stream);                                                                                        /* eam.c:250 */
         eamForce3GpuAsync(s->gpu, s->gpu.b_list, s->n_boundary_cells, s->boundary_cells, s->method, s->boundary_stream, s->spline);       /* eam.c:255 */
         cudaDeviceSynchronize();                                                                                                          /* eam.c:256 */
      } else {
         eamForce3Gpu(s->gpu,s->method, s->spline);                                                                                        /* eam.c:259 */
      }
   }
   return 0;
}

void shim_advance(SimFlat* s, real_t dt)
{
   advanceVelocityGpu(s->gpu, dt);                       /* timestep.c:138 */
   advancePositionGpu(&(s->gpu), dt);                    /* timestep.c:158 */
}

void shim_advanceVelocity(SimFlat* s, real_t dt) { advanceVelocityGpu(s->gpu, dt); }        /* timestep.c:137-141 */
void shim_advancePosition(SimFlat* s, real_t dt) { advancePositionGpu(&(s->gpu), dt); }     /* timestep.c:156-160 */

void shim_kineticEnergyGpu(SimFlat* s, real_t eLocal[2])
{
   computeEnergy(s, eLocal);                             /* timestep.c:188 */
}

void shim_redistributeAtomsGpu(SimFlat* sim)
{
   cudaMemset(sim->gpu.boxes.nAtoms + sim->boxes->nLocalBoxes, 0, (sim->boxes->nTotalBoxes - sim->boxes->nLocalBoxes) * sizeof(int));    /* timestep.c:224 */
   if (sim->usePairlist) {
      int pairlistUpdateRequired = pairlistUpdateRequiredGpu(&(sim->gpu));                                                                 /* timestep.c:228 */
      sim->gpu.genPairlist = pairlistUpdateRequired;                                                                                       /* timestep.c:229 */
      if (pairlistUpdateRequired) {
         emptyHashTableGpu(&(sim->gpu.d_hashTable));                                                                                       /* timestep.c:232 */
         updateLinkCellsGpu(sim);                                                                                                          /* timestep.c:233 */
      }
      sim->gpu.d_hashTable.nEntriesGet = 0;                                                                                                /* timestep.c:236 */
      buildAtomListGpu(sim, sim->boundary_stream);                                                                                         /* timestep.c:244 */
      if (pairlistUpdateRequired)
         sortAtomsGpu(sim, sim->boundary_stream);                                                                                          /* timestep.c:248 */
      return;
   }
   updateLinkCellsGpu(sim);                                                                                                                /* timestep.c:253 */
   if (sim->gpuAsync) {
      if (sim->method != THREAD_ATOM)
         updateNeighborsGpuAsync(sim->gpu, sim->flags, sim->gpu.boxes.nLocalBoxes - sim->n_boundary_cells, sim->interior_cells, sim->interior_stream);   /* timestep.c:260 */
      int n_interior_cells = sim->gpu.boxes.nLocalBoxes - sim->n_boundary_cells;
      eamForce1GpuAsync(sim->gpu, sim->gpu.i_list, n_interior_cells, sim->interior_cells, sim->method, sim->interior_stream, sim->spline); /* timestep.c:263 */
      eamForce2GpuAsync(sim->gpu, sim->gpu.i_list, n_interior_cells, sim->interior_cells, sim->method, sim->interior_stream, sim->spline); /* timestep.c:264 */
   }
   if (sim->atomHalo) sim->atomHalo(sim);                                                                                                  /* timestep.c:269 haloExchange(sim->atomExchange, sim) */
   buildAtomListGpu(sim, sim->boundary_stream);                                                                                            /* timestep.c:272 */
   sortAtomsGpu(sim, sim->boundary_stream);                                                                                                /* timestep.c:275 */
}

int shim_loadAtomsBuffer(void* vparms, void* data, int face, char* charBuf, real3_old shift)
{
   AtomExchangeParms* parms = (AtomExchangeParms*)vparms;
   SimFlat* sim = (SimFlat*)data;
   int nCells = parms->nCells[face];
   int* d_cellList = parms->cellListGpu[face];
   int nTotalAtomsCellList = compactCellsGpu(sim->gpu_atoms_buf, nCells, d_cellList, sim->gpu,  parms->d_natoms_buf, parms->d_partial_sums,shift,sim->boundary_stream);   /* haloExchange.c:1617 */
   cudaMemcpyAsync(charBuf, (void*)(sim->gpu_atoms_buf), nTotalAtomsCellList * sizeof(AtomMsg), cudaMemcpyDeviceToHost,sim->boundary_stream);                            /* haloExchange.c:1632 */
   cudaStreamSynchronize(sim->boundary_stream);                                                                                                                           /* haloExchange.c:1633 */
   return nTotalAtomsCellList*sizeof(AtomMsg);
}

void shim_unloadAtomsBuffer(void* data, int bufSize, char* charBuf)
{
   SimFlat* sim = (SimFlat*)data;
   int nBuf = bufSize / sizeof(AtomMsg);
   unloadAtomsBufferToGpu(charBuf, nBuf, sim, sim->gpu_atoms_buf, sim->boundary_stream);                 /* haloExchange.c:1686 */
}

int shim_loadForceBuffer(void* vparms, void* vdata, int face, char* charBuf)
{
   ForceExchangeParms* parms = (ForceExchangeParms*)vparms;
   SimFlat* s = (SimFlat*)vdata;
   int nCells = parms->nCells[face];
   int* cellListGpu = parms->sendCellsGpu[face];
   int nBuf = 0;
   loadForceBufferFromGpu(charBuf, &nBuf, nCells, cellListGpu, parms->natoms_buf[face], parms->partial_sums[face], s, s->gpu_force_buf, s->boundary_stream);   /* haloExchange.c:1872 */
   return nBuf*sizeof(ForceMsg);
}

void shim_unloadForceBuffer(void* vparms, void* vdata, int face, int bufSize, char* charBuf)
{
   ForceExchangeParms* parms = (ForceExchangeParms*)vparms;
   SimFlat* s = (SimFlat*)vdata;
   int nCells = parms->nCells[face];
   int* cellListGpu = parms->recvCellsGpu[face];
   int nBuf = bufSize / sizeof(ForceMsg);
   unloadForceBufferToGpu(charBuf, nBuf, nCells, cellListGpu, parms->natoms_buf[face], parms->partial_sums[face], s, s->gpu_force_buf, s->boundary_stream);   /* haloExchange.c:1885 */
}

/* MAXATOMS as the reference's host loops use it (timestep.c:146, haloExchange.c:1600) */
int shim_firstSlotOfCell(SimFlat* sim, int iBox) { return MAXATOMS*iBox; }
