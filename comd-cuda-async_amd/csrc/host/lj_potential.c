/* lj_potential.c -- Lennard-Jones potential plugin: parameters (ljForce.c:102-120), report (:122-134) and the
 * force entry point that BasePotential.force points at (:136-144), which hands the work to ljForceGpu.
 * The reference's LJ path never overlaps the halo exchange (it always passes cells_list = NULL, ljForce.c:141);
 * with -a 1 this version computes the interior cells on interior_stream while the atom halo exchange is in
 * flight (timestep.c redistribute hook) and only the boundary cells here -- BASELINE config 4 asks for that. */
#include "comd_host.h"
#include <stdlib.h>
#include <string.h>

static int ljForce(SimFlat* s);
static void ljPrint(FILE* file, BasePotential* pot);

static void ljDestroy(BasePotential** inppot)
{
   if (!inppot || !*inppot) return;
   free(*inppot);
   *inppot = NULL;
}

BasePotential* initLjPot(void)
{
   LjPotential* pot = (LjPotential*)calloc(1, sizeof(LjPotential));
   pot->force = ljForce;
   pot->print = ljPrint;
   pot->destroy = ljDestroy;
   pot->sigma = 2.315;                      /* Angstrom */
   pot->epsilon = 0.167;                    /* eV */
   pot->mass = 63.55 * amuToInternalMass;
   pot->lat = 3.615;
   strcpy(pot->latticeType, "FCC");
   pot->cutoff = 5 * pot->sigma;            /* this fork uses 5 sigma, not upstream CoMD's 2.5 (ljForce.c:114) */
   strcpy(pot->name, "Cu");
   pot->atomicNo = 29;
   return (BasePotential*)pot;
}

static void ljPrint(FILE* file, BasePotential* pot)
{
   LjPotential* lj = (LjPotential*)pot;
   fprintf(file, "  Potential type   : Lennard-Jones\n");
   fprintf(file, "  Species name     : %s\n", lj->name);
   fprintf(file, "  Atomic number    : %d\n", lj->atomicNo);
   fprintf(file, "  Mass             : %lg amu\n", lj->mass / amuToInternalMass);
   fprintf(file, "  Lattice Type     : %s\n", lj->latticeType);
   fprintf(file, "  Lattice spacing  : %lg Angstroms\n", lj->lat);
   fprintf(file, "  Cutoff           : %lg Angstroms\n", lj->cutoff);
   fprintf(file, "  Epsilon          : %lg eV\n", lj->epsilon);
   fprintf(file, "  Sigma            : %lg Angstroms\n", lj->sigma);
}

static int ljForce(SimFlat* sim)
{
   if (sim->gpuAsync) {
      /* interior cells were launched on interior_stream before the halo exchange */
      ensureInteriorForceLaunched(sim);
      ljForceGpuAsync(&sim->gpu, sim->n_boundary_cells, sim->gpu.boundary_cells, sim->method, sim->gpu.boundary_stream);
      comdStreamSynchronize(sim->gpu.interior_stream);
      comdStreamSynchronize(sim->gpu.boundary_stream);
   } else {
      ljForceGpu(&sim->gpu, sim->ljInterpolation, sim->gpu.boxes.nLocalBoxes, NULL, sim->pot->cutoff + sim->skinDistance, sim->method);
   }
   if (sim->usePairlist) comdPairlistGenerated(&sim->gpu);      /* every cell has been through a force call since the last build */
   return 0;
}
