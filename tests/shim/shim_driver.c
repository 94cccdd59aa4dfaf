/* shim_driver.c -- TEST INFRASTRUCTURE: whole time steps of a live simulation driven through the reference-side adapter.
 *
 * tests/test_boundary_shim.py hands this file the SimGpu of a product Simulation (atoms resident on the device) and the face cell
 * lists / periodic shifts of its single rank.  Every device-library call below goes through the scaffolds of reference_callsites.c,
 * i.e. through the reference's own call expressions and the macros / static inline adapters of include/comd_hip_shim.h:
 *   timestep.c:48-100    half kick, drift, redistribute, force, half kick            shim_advanceVelocity / shim_advancePosition
 *   timestep.c:222-276   redistributeAtomsGpu (both gpuAsync branches)               shim_redistributeAtomsGpu
 *   haloExchange.c:1493-1522 exchangeData, with the serial build's self-exchange (parallel.c:112-117: the send buffer IS what arrives):
 *                        pack -face, pack +face, unpack what the +face sent, unpack what the -face sent, through HOST buffers
 *                                                                                    shim_load/unloadAtomsBuffer, shim_load/unloadForceBuffer
 *   ljForce.c:141 / eam.c:196-264 (both gpuAsync branches)                           shim_ljForce / shim_eamForceGpu
 *   timestep.c:184-197   kineticEnergyGpu                                            shim_kineticEnergyGpu
 */
#include <stdlib.h>
#include <string.h>
#include "reference_callsites.c"

typedef struct ShimWorld {
   int   nAtomCells[6]; const int* atomCells[6];                                /* host: mkAtomCellList per face (haloExchange.c:1543-1567) */
   int   nForceCells[6]; const int* forceSend[6]; const int* forceRecv[6];      /* host: mkForceSend/RecvCellList (:1712-1801) */
   double shift[6][3];                                                          /* periodic shift of a face's outgoing atoms (:316-323) */
   int   nLocalBoxes, nTotalBoxes, method, eam, gpuAsync;
   double cutoff;
   /* filled by the driver */
   AtomExchangeParms atomParms; ForceExchangeParms forceParms;
   char *hostM, *hostP;
} ShimWorld;

static int* upload(const int* host, int n)
{
   int* d = (int*)comdDeviceMalloc((long)n * (long)sizeof(int));
   comdMemcpyHtoD(d, host, (long)n * (long)sizeof(int));
   return d;
}

static void atomHaloSelf(SimFlat* sim)                 /* haloExchange.c:1493-1522, one rank, its own neighbour on every axis */
{
   ShimWorld* w = (ShimWorld*)sim->world;
   for (int axis = 0; axis < 3; ++axis) {
      const int fM = 2 * axis, fP = fM + 1;
      real3_old sM = { (real_t)w->shift[fM][0], (real_t)w->shift[fM][1], (real_t)w->shift[fM][2] };
      real3_old sP = { (real_t)w->shift[fP][0], (real_t)w->shift[fP][1], (real_t)w->shift[fP][2] };
      const int nSendM = shim_loadAtomsBuffer(&w->atomParms, sim, fM, w->hostM, sM);
      const int nSendP = shim_loadAtomsBuffer(&w->atomParms, sim, fP, w->hostP, sP);
      /* what I sent through my minus face arrives as the message "from the plus side" and vice versa; unload -face first (:1519-1520) */
      shim_unloadAtomsBuffer(sim, nSendP, w->hostP);
      shim_unloadAtomsBuffer(sim, nSendM, w->hostM);
   }
}

static void forceHaloSelf(SimFlat* s)
{
   ShimWorld* w = (ShimWorld*)s->world;
   for (int axis = 0; axis < 3; ++axis) {
      const int fM = 2 * axis, fP = fM + 1;
      const int nSendM = shim_loadForceBuffer(&w->forceParms, s, fM, w->hostM);
      const int nSendP = shim_loadForceBuffer(&w->forceParms, s, fP, w->hostP);
      shim_unloadForceBuffer(&w->forceParms, s, fM, nSendP, w->hostP);
      shim_unloadForceBuffer(&w->forceParms, s, fP, nSendM, w->hostM);
   }
}

/* nSteps velocity-Verlet steps of timestep.c:48-100 on the simulation behind `gpu`; eOut = {ePotential, eKinetic} of the rank */
int shim_world_run(SimGpu* gpu, ShimWorld* w, int nSteps, double dt, double eOut[2])
{
   SimFlat flat; LinkCell lc; BasePotential pot;
   memset(&flat, 0, sizeof flat);
   lc.nLocalBoxes = w->nLocalBoxes; lc.nTotalBoxes = w->nTotalBoxes;
   pot.cutoff = (real_t)w->cutoff;
   flat.boxes = &lc; flat.pot = &pot;
   flat.gpu = *gpu;                                     /* the reference embeds SimGpu in SimFlat by value (CoMDTypes.h:122) */
   flat.gpu.msgBoundAtoms = 0; flat.gpu.forceScansReady = 0;
   flat.method = w->method; flat.gpuAsync = w->gpuAsync;
   flat.n_boundary_cells = gpu->n_boundary_cells;
   flat.boundary_cells = gpu->boundary_cells; flat.interior_cells = gpu->interior_cells;
   flat.boundary_stream = gpu->boundary_stream; flat.interior_stream = gpu->interior_stream;
   flat.atomHalo = atomHaloSelf; flat.forceHalo = forceHaloSelf; flat.world = w;

   int maxCells = 1;
   for (int f = 0; f < 6; ++f) {
      w->atomParms.nCells[f] = w->nAtomCells[f]; w->atomParms.cellListGpu[f] = upload(w->atomCells[f], w->nAtomCells[f]);
      w->forceParms.nCells[f] = w->nForceCells[f];
      w->forceParms.sendCellsGpu[f] = upload(w->forceSend[f], w->nForceCells[f]);
      w->forceParms.recvCellsGpu[f] = upload(w->forceRecv[f], w->nForceCells[f]);
      if (w->nAtomCells[f] > maxCells) maxCells = w->nAtomCells[f];
   }
   const long maxAtomsMsg = (long)maxCells * gpu->maxAtoms;         /* bufCapacity of haloExchange.c:202-207 */
   flat.gpu_atoms_buf = comdShimAtomsBufAlloc(maxAtomsMsg * (long)sizeof(AtomMsg));
   flat.gpu_force_buf = (char*)comdDeviceMalloc(maxAtomsMsg * (long)sizeof(ForceMsg));
   w->hostM = (char*)malloc((size_t)maxAtomsMsg * sizeof(AtomMsg)); w->hostP = (char*)malloc((size_t)maxAtomsMsg * sizeof(AtomMsg));

   for (int step = 0; step < nSteps; ++step) {          /* timestep.c:52-95 */
      shim_advanceVelocity(&flat, (real_t)(0.5 * dt));
      shim_advancePosition(&flat, (real_t)dt);
      shim_redistributeAtomsGpu(&flat);
      if (w->eam) shim_eamForceGpu(&flat); else shim_ljForce(&flat);
      shim_advanceVelocity(&flat, (real_t)(0.5 * dt));
   }
   real_t e[2];
   shim_kineticEnergyGpu(&flat, e);
   eOut[0] = e[0]; eOut[1] = e[1];

   comdDeviceSynchronize();
   free(w->hostM); free(w->hostP);
   comdShimAtomsBufFree(flat.gpu_atoms_buf); comdDeviceFree(flat.gpu_force_buf);
   for (int f = 0; f < 6; ++f) { comdDeviceFree(w->atomParms.cellListGpu[f]); comdDeviceFree(w->forceParms.sendCellsGpu[f]); comdDeviceFree(w->forceParms.recvCellsGpu[f]); }
   *gpu = flat.gpu;                                     /* lazily allocated members (rows, lists, the adapter's scan scratch) belong to the simulation */
   return 0;
}
