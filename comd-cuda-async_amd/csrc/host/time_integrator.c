/* time_integrator.c -- velocity-Verlet driver: half kick, drift, redistribute, force, half kick
 * (timestep.c:48-100), computeForce through the potential's function pointer (:102-105), the energy
 * reduction (:184-197) and the per-step atom redistribution (:222-276).  Every phase is a call into
 * include/comd_hip.h; the only blocking device round-trip per step on one rank is none at all
 * (the reference has >= 20) -- energies are fetched every printRate steps. */
#include "comd_host.h"
#include <stdlib.h>

static void advanceVelocity(SimFlat* s, real_t dt) { advanceVelocityGpu(&s->gpu, dt); }

double timestep(SimFlat* s, int nSteps, real_t dt)
{
   for (int ii = 0; ii < nSteps; ++ii) {
      /* half kick + drift fused in one kernel (same arithmetic as advanceVelocity(dt/2) then advancePosition(dt)); from the second step
       * on they ride with the previous step's closing half kick (below) */
      if (ii == 0) {
         startTimer(velocityTimer);
         startTimer(positionTimer);
         advanceVelocityPositionGpu(&s->gpu, 0.5 * dt, dt);
         stopTimer(positionTimer);
         stopTimer(velocityTimer);
      }

      /* e[] is read by kineticEnergyGpu below only: the last step's forces must carry energies, the others need not
       * (set before redistributeAtoms: with -a 1 it already launches the interior cells' force work) */
      comdSetEnergyNeeded(&s->gpu, ii == nSteps - 1);
      startTimer(redistributeTimer);
      redistributeAtoms(s);
      stopTimer(redistributeTimer);

      startTimer(computeForceTimer);
      computeForce(s);
      stopTimer(computeForceTimer);
      comdSetEnergyNeeded(&s->gpu, 1);
      /* device status words (cell overflow, lost atom, message overflow, row overflow) of the step BEFORE, without a synchronisation: a run with a
       * dropped atom stops a step or two later (round 2: at the next energy read, up to printRate steps on) */
      comdPollStatus(&s->gpu, s->gpu.boundary_stream, "timestep");

      startTimer(velocityTimer);
      if (ii == nSteps - 1) advanceVelocity(s, 0.5 * dt);
      else {
         /* closing half kick of this step + opening half kick and drift of the next: one pass over the atoms, the two kicks still two
          * separate roundings (bit-identical to the three calls of timestep.c:52-58, 95) */
         startTimer(positionTimer);
         advanceVelocityVelocityPositionGpu(&s->gpu, 0.5 * dt, 0.5 * dt, dt);
         stopTimer(positionTimer);
      }
      stopTimer(velocityTimer);
   }
   kineticEnergyGpu(s);
   return s->ePotential;
}

void computeForce(SimFlat* s) { s->pot->force(s); }

void kineticEnergyGpu(SimFlat* s)
{
   real_t eLocal[2], eSum[2];
   computeEnergy(&s->gpu, eLocal);
   comdCheckStatus(&s->gpu, "kineticEnergyGpu");      /* cell / message overflow and lost atoms surface here */
   if (s->gpu.do_eam) {
      /* cta_cell, thread_atom: bricks whose block outgrew the LDS image took the thread-per-atom form (correct, many times slower).  The image was sized for the occupancies of
       * the first launch; a system that has changed since (heating, a density front) is noticed here, once per energy read, and the next launch sizes it again */
      int st[3];
      comdEamBrickStats(&s->gpu, st);
      if (st[0] > 0 && (s->method == CTA_CELL || s->method == THREAD_ATOM || s->method == WARP_ATOM)) {
         if (!s->quiet && printRank()) fprintf(stderr, "eamForce: %d brick launches took the thread-per-atom form since the last energy read (image of %d records): re-sizing the image\n", st[0], st[2]);
         comdEamBrickResize(&s->gpu);
      }
   }
   startTimer(commReduceTimer);
   addRealParallel(eLocal, eSum, 2);
   stopTimer(commReduceTimer);
   s->ePotential = eSum[0];
   s->eKinetic = eSum[1];
}

static void launchInteriorForce(SimFlat* sim)
{
   SimGpu* g = &sim->gpu;
   if (sim->gpu.do_eam) {
      eamForce1GpuAsync(g, g->n_interior_cells, g->interior_cells, sim->method, g->interior_stream, sim->spline);
      eamForce2GpuAsync(g, g->n_interior_cells, g->interior_cells, sim->method, g->interior_stream, sim->spline);
   } else {
      ljForceGpuAsync(g, g->n_interior_cells, g->interior_cells, sim->method, g->interior_stream);
   }
   sim->interiorLaunched = 1;
}

/* -a 1: the potentials call this before their boundary work; starts the interior cells if redistributeAtoms could not
 * (a Verlet-list build had to come first) */
void ensureInteriorForceLaunched(SimFlat* sim)
{
   if (sim->gpuAsync && !sim->interiorLaunched) {
      comdStreamSynchronize(sim->gpu.boundary_stream);
      launchInteriorForce(sim);
   }
}

static void redistributeAtomsCells(SimFlat* sim, int overlapInterior);

/* timestep.c:278-352 redistributeAtomsGpuNL.  Lists valid (no atom has moved more than skin/2 since the build, on any rank):
 * nothing is re-binned, atoms keep their slots, only the halo copies' positions are refreshed.  Otherwise: the ordinary
 * redistribution (re-bin, migrate, full atom exchange, gid sort) followed by a list build.  The reference also rebuilds on every
 * rank whenever an atom changed owner (:331-350); here owners only change inside the ordinary redistribution. */
static void redistributeAtomsNL(SimFlat* sim)
{
   SimGpu* g = &sim->gpu;
   /* The reference's rule (gpu_kernels.cu:1449-1484): has an atom moved more than skin/2 since the build?  Asked of the drift kernel that has just run -- a stream
    * synchronisation per step (35 us of device idle at EAM 80^3).  COMD_NL_DEFERRED=1: the form that never drains the stream (comd_hip.h
    * comdNeighborListUpdateDeferredGpu: decided two drifts late against a threshold lowered by what two steps can add) -- measured at -1 % of the step, because the
    * lower threshold rebuilds the lists every 18 instead of every 22 steps; the blocking form stays the default. */
   static int deferred = -1;
   if (deferred < 0) { const char* e = getenv("COMD_NL_DEFERRED"); deferred = e && atoi(e) != 0; }
   int need = deferred ? comdNeighborListUpdateDeferredGpu(g) : neighborListUpdateRequiredGpu(g), needAll = 0;
   startTimer(commReduceTimer);
   maxIntParallel(&need, &needAll, 1);
   stopTimer(commReduceTimer);
   sim->interiorLaunched = 0;
   if (needAll) {
      /* many steps since the last full exchange: the message counts may have drifted past the agreed slack -- swap exact sizes once */
      invalidateHaloSizes(sim->atomExchange);
      if (sim->gpu.do_eam) invalidateHaloSizes(((EamPotential*)sim->pot)->forceExchange);
      redistributeAtomsCells(sim, 0);
      startTimer(neighborListBuildTimer);
      buildNeighborListGpu(g, sim->method, 0);
      preparePositionExchange(sim->positionExchange, sim);
      stopTimer(neighborListBuildTimer);
      sim->nlBuilds++;
   } else {
      if (sim->gpuAsync) { comdStreamSynchronize(g->boundary_stream); launchInteriorForce(sim); }
      startTimer(atomHaloTimer);
      haloExchange(sim->positionExchange, sim);
      stopTimer(atomHaloTimer);
   }
}

void redistributeAtoms(SimFlat* sim)
{
   if (sim->useNL) redistributeAtomsNL(sim);
   else { sim->interiorLaunched = 0; redistributeAtomsCells(sim, 1); }
}

/* timestep.c:222-276 redistributeAtomsGpu */
static void redistributeAtomsCells(SimFlat* sim, int overlapInterior)
{
   SimGpu* g = &sim->gpu;
   /* empties the halo cells, moves atoms that left their cell, compacts + gid-sorts the cells that changed -- the last only when something reads the cells before
    * sortAtomsGpu does it again: a pack kernel (an axis with a real peer) or the interior force launch of the overlap mode */
   g->skipSortAfterUpdate = haloMirrorFirstAxis(sim->atomExchange) == 0 && !(sim->gpuAsync && overlapInterior);
   updateLinkCellsGpu(g, g->boundary_stream);

   if (sim->gpuAsync && overlapInterior) {
      /* local cells are final: start the interior force work while the halo exchange runs (timestep.c:257-265) */
      comdStreamSynchronize(g->boundary_stream);
      launchInteriorForce(sim);
   }

   startTimer(atomHaloTimer);
   haloExchange(sim->atomExchange, sim);
   stopTimer(atomHaloTimer);

   buildAtomListGpu(g, g->boundary_stream);
   sortAtomsGpu(g, g->boundary_stream);          /* halo cells and cells that received migrants */
}
