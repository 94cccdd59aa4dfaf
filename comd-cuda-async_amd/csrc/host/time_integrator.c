/* time_integrator.c -- velocity-Verlet driver: half kick, drift, redistribute, force, half kick
 * (timestep.c:48-100), computeForce through the potential's function pointer (:102-105), the energy
 * reduction (:184-197) and the per-step atom redistribution (:222-276).  Every phase is a call into
 * include/comd_hip.h; the only blocking device round-trip per step on one rank is none at all
 * (the reference has >= 20) -- energies are fetched every printRate steps. */
#include "comd_host.h"

static void advanceVelocity(SimFlat* s, real_t dt) { advanceVelocityGpu(&s->gpu, dt); }

double timestep(SimFlat* s, int nSteps, real_t dt)
{
   for (int ii = 0; ii < nSteps; ++ii) {
      /* half kick + drift fused in one kernel (same arithmetic as advanceVelocity(dt/2) then advancePosition(dt)) */
      startTimer(velocityTimer);
      startTimer(positionTimer);
      advanceVelocityPositionGpu(&s->gpu, 0.5 * dt, dt);
      stopTimer(positionTimer);
      stopTimer(velocityTimer);

      /* e[] is read by kineticEnergyGpu below only: the last step's forces must carry energies, the others need not
       * (set before redistributeAtoms: with -a 1 it already launches the interior cells' force work) */
      comdSetEnergyNeeded(ii == nSteps - 1);
      startTimer(redistributeTimer);
      redistributeAtoms(s);
      stopTimer(redistributeTimer);

      startTimer(computeForceTimer);
      computeForce(s);
      stopTimer(computeForceTimer);
      comdSetEnergyNeeded(1);

      startTimer(velocityTimer);
      advanceVelocity(s, 0.5 * dt);
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
   startTimer(commReduceTimer);
   addRealParallel(eLocal, eSum, 2);
   stopTimer(commReduceTimer);
   s->ePotential = eSum[0];
   s->eKinetic = eSum[1];
}

/* timestep.c:222-276 redistributeAtomsGpu */
void redistributeAtoms(SimFlat* sim)
{
   SimGpu* g = &sim->gpu;
   /* empties the halo cells, moves atoms that left their cell, compacts + gid-sorts the cells that changed */
   updateLinkCellsGpu(g, g->boundary_stream);

   if (sim->gpuAsync) {
      /* local cells are final: start the interior force work while the halo exchange runs (timestep.c:257-265) */
      comdStreamSynchronize(g->boundary_stream);
      if (sim->gpu.do_eam) {
         eamForce1GpuAsync(g, g->n_interior_cells, g->interior_cells, sim->method, g->interior_stream, sim->spline);
         eamForce2GpuAsync(g, g->n_interior_cells, g->interior_cells, sim->method, g->interior_stream, sim->spline);
      } else {
         ljForceGpuAsync(g, g->n_interior_cells, g->interior_cells, sim->method, g->interior_stream);
      }
   }

   startTimer(atomHaloTimer);
   haloExchange(sim->atomExchange, sim);
   stopTimer(atomHaloTimer);

   buildAtomListGpu(g, g->boundary_stream);
   sortAtomsGpu(g, g->boundary_stream);          /* halo cells and cells that received migrants */
}
