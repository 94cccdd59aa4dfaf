/* simulation.c -- simulation set-up/tear-down, the stdout report and the driver loop.
 * Follows CoMD.c: main (:86-187), initSimulation (:200-327), destroySimulation (:330-351), initValidate/validateResult
 * (:395-440), sumAtoms (:442-457), printThings (:463-494), printSimulationDataYaml (:498-552), sanityChecks (:555-604).
 * Table headers and number formats are the reference's, verbatim, because downstream scripts parse them. */
#include "comd_host.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

static BasePotential* initPotential(int doeam, const char* potDir, const char* potName, const char* potType)
{
   return doeam ? initEamPot(potDir, potName, potType) : initLjPot();
}

static SpeciesData* initSpecies(BasePotential* pot)
{
   SpeciesData* species = (SpeciesData*)calloc(1, sizeof(SpeciesData));
   strcpy(species->name, pot->name);
   species->atomicNo = pot->atomicNo;
   species->mass = pot->mass;
   return species;
}

static void sanityChecks(Command cmd, double cutoff, double latticeConst, char latticeType[8])
{
   int failCode = 0;
   if (cmd.xproc * cmd.yproc * cmd.zproc != getNRanks()) {
      failCode |= 1;
      if (printRank()) fprintf(screenOut, "\nNumber of MPI ranks must match xproc * yproc * zproc\n");
   }
   double minx = 2 * cutoff * cmd.xproc, miny = 2 * cutoff * cmd.yproc, minz = 2 * cutoff * cmd.zproc;
   double sizex = cmd.nx * latticeConst, sizey = cmd.ny * latticeConst, sizez = cmd.nz * latticeConst;
   if (sizex < minx || sizey < miny || sizez < minz) {
      failCode |= 2;
      if (printRank())
         fprintf(screenOut, "\nSimulation too small.\n"
                 "  Increase the number of unit cells to make the simulation\n"
                 "  at least (%3.2f, %3.2f. %3.2f) Ansgstroms in size\n", minx, miny, minz);
   }
   if (strcasecmp(latticeType, "FCC") != 0) {
      failCode |= 4;
      if (printRank()) fprintf(screenOut, "\nOnly FCC Lattice type supported, not %s. Fatal Error.\n", latticeType);
   }
   if (failCode != 0) exit(failCode);
}

/* link-cell slot capacity when the user gives none: the lattice's largest occupancy plus head-room for thermal
 * motion and the initial displacement, rounded so that waves never straddle cells (LJ: multiple of 64) or cells
 * never straddle waves (EAM: power of two). */
static int chooseMaxAtoms(int latticeMax, real_t delta, const LinkCell* boxes, int doeam, int useNL)
{
   real_t minBox = fmin(boxes->boxSize[0], fmin(boxes->boxSize[1], boxes->boxSize[2]));
   int want = (int)ceil(latticeMax * (1.10 + 3.0 * delta / minBox)) + 8;
   if (!doeam) return ((want + 63) / 64) * 64;
   /* the EAM kernels take any capacity: a multiple of 4 (16-byte rows of ints).  Cells of cutoff (+ skin) hold 4..32 atoms of a
    * perfect lattice: 20 or 44 slots instead of 32 or 64 is a third less for every slot-wise kernel and for the atom arrays */
   (void)useNL;
   return ((want + 3) / 4) * 4;
}

static int cmpInt(const void* a, const void* b) { return (*(const int*)a > *(const int*)b) - (*(const int*)a < *(const int*)b); }

/* gpu_utility.c:73-163 SetBoundaryCells, host half: ring 1 = local cells that touch the halo, ring 2 = their local
 * neighbours; "boundary" = ring 1 + ring 2, "interior" = the rest. */
void setBoundaryCellsHost(SimFlat* sim, HaloExchange* hh)
{
   (void)hh;
   LinkCell* b = sim->boxes;
   const int n = b->nLocalBoxes;
   int* type = (int*)calloc((size_t)n, sizeof(int));
   sim->boundary1_cells_h = (int*)malloc((size_t)n * sizeof(int));
   sim->boundary_cells_h = (int*)malloc((size_t)n * sizeof(int));
   sim->interior_cells_h = (int*)malloc((size_t)n * sizeof(int));
   int n1 = 0, nb = 0, ni = 0, nbr[27];
   for (int i = 0; i < n; ++i) {
      int ix, iy, iz; comdTupleFromBox(&b->geom, i, &ix, &iy, &iz);
      if (ix == 0 || iy == 0 || iz == 0 || ix == b->gridSize[0] - 1 || iy == b->gridSize[1] - 1 || iz == b->gridSize[2] - 1) {
         type[i] = 1; sim->boundary1_cells_h[n1++] = i; sim->boundary_cells_h[nb++] = i;
      }
   }
   for (int i = 0; i < n; ++i) {
      if (type[i]) continue;
      getNeighborBoxes(b, i, nbr);
      for (int j = 0; j < 27; ++j) if (nbr[j] < n && type[nbr[j]] == 1) { type[i] = 2; sim->boundary_cells_h[nb++] = i; break; }
   }
   for (int i = 0; i < n; ++i) if (type[i] == 0) sim->interior_cells_h[ni++] = i;
   /* both rings in one ascending run: list neighbours are then spatial neighbours, which is what the kernels' contiguous
    * per-XCD ranges and the scalar cache want */
   qsort(sim->boundary_cells_h, (size_t)nb, sizeof(int), cmpInt);
   sim->n_boundary1_cells = n1; sim->n_boundary_cells = nb;
   free(type);
   SetBoundaryCells(&sim->gpu, nb, sim->boundary_cells_h, ni, sim->interior_cells_h, n1, sim->boundary1_cells_h);
}

/* host half of initSimulation: everything up to (not including) device allocation.  Needs no GPU, which is what
 * lets the CPU test-suite check lattice, momenta, link cells and halo cell lists against the oracle. */
SimFlat* initSimulationHost(Command cmd)
{
   SimFlat* sim = (SimFlat*)calloc(1, sizeof(SimFlat));
   sim->cmdDoeam = cmd.doeam;
   sim->nSteps = cmd.nSteps; sim->printRate = cmd.printRate; sim->dt = cmd.dt;
   sim->gpuAsync = cmd.gpuAsync; sim->gpuProfile = cmd.gpuProfile;
   sim->quiet = cmd.quiet; sim->iStepPrev = -1; sim->firstPrint = 1;
   if (sim->gpuProfile) sim->nSteps = 0;

   if (!strcmp(cmd.method, "thread_atom")) sim->method = THREAD_ATOM;
   else if (!strcmp(cmd.method, "cta_cell")) sim->method = CTA_CELL;
   else if (!strcmp(cmd.method, "thread_atom_nl")) sim->method = THREAD_ATOM_NL;
   else if (!strcmp(cmd.method, "warp_atom_nl")) {
      if (printRank()) printf("Method warp_atom_nl runs as thread_atom_nl in this build.\n");
      sim->method = THREAD_ATOM_NL;
   }
   else if (!strcmp(cmd.method, "warp_atom")) {
      if (printRank()) printf("Method warp_atom runs as thread_atom in this build.\n");
      sim->method = THREAD_ATOM;
   }
   else { printf("Error: You have to specify a valid method: -m [thread_atom,thread_atom_nl,cta_cell]\n"); exit(-1); }
   sim->useNL = sim->method == THREAD_ATOM_NL;
   if (cmd.ljInterpolation) {
      printf("Error: -I is outside this build's scope (SURVEY.md section 8f).\n"); exit(-1);
   }
   /* -P (CoMD.c:271, gpu_utility.c:474-500): cubic-spline tables in r^2 for phi and rho, for every force method (gpu_kernels.cu:164-226) */
   sim->spline = cmd.spline;
   if (sim->spline && !cmd.doeam) { printf("Error: -P applies to EAM (-e).\n"); exit(-1); }
   /* -L (CoMD.c:250-255, ljForce.c:141): pairlist bits for the CTA-per-cell LJ kernel; same skin, cells and rebuild rule as the lists */
   sim->usePairlist = cmd.usePairlist;
   if (sim->usePairlist && (cmd.doeam || sim->method != CTA_CELL)) { printf("Error: -L applies to LJ with -m cta_cell.\n"); exit(-1); }
   sim->useNL = sim->useNL || sim->usePairlist;        /* frozen slots between rebuilds, positional halo refresh */

   sim->pot = initPotential(cmd.doeam, cmd.potDir, cmd.potName, cmd.potType);
   if (!cmd.doeam && cmd.ljCutoffSigmas > 0.0) sim->pot->cutoff = cmd.ljCutoffSigmas * ((LjPotential*)sim->pot)->sigma;
   if (sim->spline) eamUseSplines(sim->pot);
   real_t latticeConstant = cmd.lat;
   if (cmd.lat < 0.0) latticeConstant = sim->pot->lat;
   sanityChecks(cmd, sim->pot->cutoff, latticeConstant, sim->pot->latticeType);
   sim->species = initSpecies(sim->pot);

   const real_t globalExtent[3] = { cmd.nx * latticeConstant, cmd.ny * latticeConstant, cmd.nz * latticeConstant };
   sim->domain = initDecomposition(cmd.xproc, cmd.yproc, cmd.zproc, globalExtent);
   /* CoMD.c:257-268: the *_nl methods list neighbours out to cutoff + skin and size the link cells to match */
   sim->skinDistance = sim->useNL ? sim->pot->cutoff * cmd.relativeSkinDistance : 0.0;
   if (sim->useNL && printRank() && !cmd.quiet) printf("Skin-Distance: %f\n", sim->skinDistance);
   if (sim->useNL && sim->skinDistance <= 0.0) { printf("Error: the *_nl methods need a positive skin distance (-S)\n"); exit(-1); }
   sim->boxes = initLinkCells(sim->domain, sim->pot->cutoff + sim->skinDistance, cmd.doHilbert);

   int cap = cmd.maxAtoms;
   if (cap <= 0) {
      int localMax = countFccLattice(cmd.nx, cmd.ny, cmd.nz, latticeConstant, sim->domain, sim->boxes), globalMaxOcc;
      maxIntParallel(&localMax, &globalMaxOcc, 1);
      cap = chooseMaxAtoms(globalMaxOcc, cmd.initialDelta, sim->boxes, cmd.doeam, sim->useNL);
   }
   sim->boxes->maxAtoms = cap;
   sim->atoms = initAtoms(sim->boxes);

   createFccLattice(cmd.nx, cmd.ny, cmd.nz, latticeConstant, sim);
   setTemperature(sim, cmd.temperature);
   randomDisplacements(sim, cmd.initialDelta);
   return sim;
}

SimFlat* initSimulation(Command cmd)
{
   SimFlat* sim = initSimulationHost(cmd);
   const int cap = sim->boxes->maxAtoms;

   /* device state: AllocateGpu / CopyDataToGpu (gpu_utility.c:165-282, 432-600) */
   GpuConfig cfg; memset(&cfg, 0, sizeof cfg);
   cfg.maxAtoms = cap; cfg.nLocalBoxes = sim->boxes->nLocalBoxes; cfg.nTotalBoxes = sim->boxes->nTotalBoxes;
   for (int a = 0; a < 3; ++a) {
      cfg.gridSize[a] = sim->boxes->gridSize[a]; cfg.localMin[a] = sim->boxes->localMin[a];
      cfg.localMax[a] = sim->boxes->localMax[a]; cfg.boxSize[a] = sim->boxes->boxSize[a];
   }
   cfg.do_eam = cmd.doeam; cfg.gpuAsync = cmd.gpuAsync; cfg.rank = getMyRank(); cfg.mass = sim->species[0].mass;
   if (cmd.doeam) {
      EamPotential* e = (EamPotential*)sim->pot;
      cfg.eamCutoff = e->cutoff;
      cfg.nPhi = e->phi->n; cfg.phiX0 = e->phi->x0; cfg.phiInvDx = e->phi->invDx; cfg.phiValues = e->phi->values - 1;
      cfg.nRho = e->rho->n; cfg.rhoX0 = e->rho->x0; cfg.rhoInvDx = e->rho->invDx; cfg.rhoValues = e->rho->values - 1;
      cfg.nF = e->f->n;     cfg.fX0 = e->f->x0;     cfg.fInvDx = e->f->invDx;     cfg.fValues = e->f->values - 1;
      cfg.phiSpline = e->phiSpline; cfg.rhoSpline = e->rhoSpline;
   } else {
      LjPotential* lj = (LjPotential*)sim->pot;
      cfg.ljCutoff = lj->cutoff; cfg.ljSigma = lj->sigma; cfg.ljEpsilon = lj->epsilon;
   }
   int* nbrTable = (int*)malloc((size_t)sim->boxes->nLocalBoxes * 27 * sizeof(int));
   for (int iBox = 0; iBox < sim->boxes->nLocalBoxes; ++iBox) {       /* self first (gpu_utility.c:520-531) */
      int nbr[27], c = 0;
      getNeighborBoxes(sim->boxes, iBox, nbr);
      nbrTable[iBox * 27 + c++] = iBox;
      for (int j = 0; j < 27; ++j) if (nbr[j] != iBox) nbrTable[iBox * 27 + c++] = nbr[j];
   }
   cfg.neighborCells = nbrTable;
   cfg.boxIDLookUp = sim->boxes->boxIDLookUp; cfg.boxIDLookUpReverse = sim->boxes->boxIDLookUpReverse;
   cfg.skinDistance = sim->skinDistance; cfg.maxNeighbors = cmd.maxNeighbors; cfg.usePairlist = sim->usePairlist;
   cfg.latticeConstant = cmd.lat < 0.0 ? sim->pot->lat : cmd.lat;
   AllocateGpu(&sim->gpu, &cfg);
   free(nbrTable);
   /* cta_cell: pass 1 applies the embedding function to the atoms it has just summed rhobar for; eamForce2Gpu then has nothing left to do.
    * Not in profiling mode (-s), which runs pass 1 alone and expects the reference's pass-1 state. */
   sim->gpu.fuseEmbed = cmd.doeam && (sim->method == CTA_CELL || sim->method == THREAD_ATOM_NL || sim->method == THREAD_ATOM || sim->method == WARP_ATOM) && !sim->gpuProfile;   /* (the list method: when the device library keeps brick rows, comd_hip.h NeighborListGpu.slabFormat 4) */

   sim->atomExchange = initAtomHaloExchange(sim->domain, sim->boxes, 1);
   if (cmd.doeam) ((EamPotential*)sim->pot)->forceExchange = initForceHaloExchange(sim->domain, sim->boxes, 1);
   if (sim->useNL) sim->positionExchange = initPositionHaloExchange(sim->domain, sim->boxes, 1);
   setBoundaryCellsHost(sim, sim->atomExchange);
   CopyDataToGpu(&sim->gpu, &sim->atoms->h);

   /* forces must exist before the first half kick (CoMD.c:303-320) */
   if (!sim->gpuProfile) {
      startTimer(redistributeTimer);
      redistributeAtoms(sim);
      stopTimer(redistributeTimer);
   }
   startTimer(computeForceTimer);
   computeForce(sim);
   stopTimer(computeForceTimer);
   kineticEnergyGpu(sim);
   return sim;
}

void destroySimulation(SimFlat** ps)
{
   if (!ps || !*ps) return;
   SimFlat* s = *ps;
   if (s->atomExchange) destroyHaloExchange(&s->atomExchange);
   if (s->positionExchange) destroyHaloExchange(&s->positionExchange);
   BasePotential* pot = s->pot;
   if (pot) pot->destroy(&pot);
   if (s->gpu.boxes.nAtoms) DestroyGpu(&s->gpu);
   destroyLinkCells(&s->boxes);
   destroyAtoms(s->atoms);
   free(s->boundary_cells_h); free(s->interior_cells_h); free(s->boundary1_cells_h);
   free(s->species); free(s->domain); free(s);
   *ps = NULL;
}

void sumAtoms(SimFlat* s)
{
   updateNAtomsCpu(&s->gpu, s->boxes->nAtoms);     /* the reference forgets this refresh (CoMD.c:445-446) */
   s->atoms->nLocal = 0;
   for (int i = 0; i < s->boxes->nLocalBoxes; i++) s->atoms->nLocal += s->boxes->nAtoms[i];
   startTimer(commReduceTimer);
   addIntParallel(&s->atoms->nLocal, &s->atoms->nGlobal, 1);
   stopTimer(commReduceTimer);
}

void printThings(SimFlat* s, int iStep, double elapsedTime)
{
   int nEval = iStep - s->iStepPrev;               /* 1 for the zeroth step */
   s->iStepPrev = iStep;
   if (!printRank() || s->quiet) return;
   if (s->firstPrint) {
      s->firstPrint = 0;
      fprintf(screenOut,
              "#                                                                                         Performance\n"
              "#  Loop   Time(fs)       Total Energy   Potential Energy     Kinetic Energy  Temperature   (us/atom)     # Atoms\n");
      fflush(screenOut);
   }
   real_t time = iStep * s->dt;
   real_t eTotal = (s->ePotential + s->eKinetic) / s->atoms->nGlobal;
   real_t eK = s->eKinetic / s->atoms->nGlobal;
   real_t eU = s->ePotential / s->atoms->nGlobal;
   real_t Temp = (s->eKinetic / s->atoms->nGlobal) / (kB_eV * 1.5);
   double timePerAtom = 1.0e6 * elapsedTime / (double)(nEval * s->atoms->nLocal);
   fprintf(screenOut, " %6d %10.2f %18.12f %18.12f %18.12f %12.4f %10.4f %12d\n",
           iStep, time, eTotal, eU, eK, Temp, timePerAtom, s->atoms->nGlobal);
}

Validate* initValidate(SimFlat* sim)
{
   sumAtoms(sim);
   Validate* val = (Validate*)calloc(1, sizeof(Validate));
   val->eTot0 = (sim->ePotential + sim->eKinetic) / sim->atoms->nGlobal;
   val->nAtoms0 = sim->atoms->nGlobal;
   if (printRank() && !sim->quiet) {
      fprintf(screenOut, "\n");
      printSeparator(screenOut);
      fprintf(screenOut, "Initial energy : %14.12f, atom count : %d \n", val->eTot0, val->nAtoms0);
      fprintf(screenOut, "\n");
   }
   return val;
}

void validateResult(const Validate* val, SimFlat* sim)
{
   if (!printRank() || sim->quiet) return;
   real_t eFinal = (sim->ePotential + sim->eKinetic) / sim->atoms->nGlobal;
   int nAtomsDelta = (sim->atoms->nGlobal - val->nAtoms0);
   fprintf(screenOut, "\n\n");
   fprintf(screenOut, "Simulation Validation:\n");
   fprintf(screenOut, "  Initial energy  : %14.12f\n", val->eTot0);
   fprintf(screenOut, "  Final energy    : %14.12f\n", eFinal);
   fprintf(screenOut, "  eFinal/eInitial : %f\n", eFinal / val->eTot0);
   if (nAtomsDelta == 0) fprintf(screenOut, "  Final atom count : %d, no atoms lost\n", sim->atoms->nGlobal);
   else {
      fprintf(screenOut, "#############################\n");
      fprintf(screenOut, "# WARNING: %6d atoms lost #\n", nAtomsDelta);
      fprintf(screenOut, "#############################\n");
   }
}

void printSimulationDataYaml(FILE* file, SimFlat* s)
{
   int maxOcc = maxOccupancy(s->boxes);
   if (!printRank() || !file) return;
   fprintf(file, "Simulation data: \n");
   fprintf(file, "  Total atoms        : %d\n", s->atoms->nGlobal);
   fprintf(file, "  Min global bounds  : [ %14.10f, %14.10f, %14.10f ]\n", s->domain->globalMin[0], s->domain->globalMin[1], s->domain->globalMin[2]);
   fprintf(file, "  Max global bounds  : [ %14.10f, %14.10f, %14.10f ]\n", s->domain->globalMax[0], s->domain->globalMax[1], s->domain->globalMax[2]);
   printSeparator(file);
   fprintf(file, "Decomposition data: \n");
   fprintf(file, "  Processors         : %6d,%6d,%6d\n", s->domain->procGrid[0], s->domain->procGrid[1], s->domain->procGrid[2]);
   fprintf(file, "  Local boxes        : %6d,%6d,%6d = %8d\n", s->boxes->gridSize[0], s->boxes->gridSize[1], s->boxes->gridSize[2],
           s->boxes->gridSize[0] * s->boxes->gridSize[1] * s->boxes->gridSize[2]);
   fprintf(file, "  Box size           : [ %14.10f, %14.10f, %14.10f ]\n", s->boxes->boxSize[0], s->boxes->boxSize[1], s->boxes->boxSize[2]);
   fprintf(file, "  Box factor         : [ %14.10f, %14.10f, %14.10f ] \n", s->boxes->boxSize[0] / s->pot->cutoff,
           s->boxes->boxSize[1] / s->pot->cutoff, s->boxes->boxSize[2] / s->pot->cutoff);
   fprintf(file, "  Max Link Cell Occupancy: %d of %d\n", maxOcc, s->boxes->maxAtoms);
   printSeparator(file);
   fprintf(file, "Potential data: \n");
   s->pot->print(file, s->pot);
   int perAtomSize = 10 * sizeof(real_t) + 2 * sizeof(int);
   float totalMemLocal = (float)perAtomSize * s->atoms->nLocal / 1024 / 1024;
   float totalMemGlobal = (float)perAtomSize * s->atoms->nGlobal / 1024 / 1024;
   float paddedMemLocal = (float)s->boxes->nLocalBoxes * ((float)perAtomSize * s->boxes->maxAtoms) / 1024 / 1024;
   float paddedMemTotal = (float)s->boxes->nTotalBoxes * ((float)perAtomSize * s->boxes->maxAtoms) / 1024 / 1024;
   printSeparator(file);
   fprintf(file, "Memory data: \n");
   fprintf(file, "  Intrinsic atom footprint = %4d B/atom \n", perAtomSize);
   fprintf(file, "  Total atom footprint     = %7.3f MB (%6.2f MB/node)\n", totalMemGlobal, totalMemLocal);
   fprintf(file, "  Link cell atom footprint = %7.3f MB/node\n", paddedMemLocal);
   fprintf(file, "  Link cell atom footprint = %7.3f MB/node (including halo cell data\n", paddedMemTotal);
   fflush(file);
}

/* ---- the reference's main(), as a callable (CoMD.c:86-187) -------------------------------------------------- */
int comdMain(int argc, char** argv)
{
   profileStart(totalTimer);
   yamlBegin();
   timestampBarrier("Starting Initialization\n");
   yamlAppInfo(yamlFile);
   yamlAppInfo(screenOut);
   Command cmd = parseCommandLine(argc, argv);
   printCmdYaml(yamlFile, &cmd);
   printCmdYaml(screenOut, &cmd);

   SimFlat* sim = initSimulation(cmd);
   if (cmd.deviceTimers || (getenv("COMD_DEVICE_TIMERS") && atoi(getenv("COMD_DEVICE_TIMERS")) != 0))
      timersUseDevice(1, sim->gpu.boundary_stream, cmd.doeam ? 176.0 * sizeof(real_t) / 8.0 : 56.0 * sizeof(real_t) / 8.0);      /* force bytes per atom */
   sumAtoms(sim);
   printSimulationDataYaml(yamlFile, sim);
   printSimulationDataYaml(screenOut, sim);
   Validate* validate = initValidate(sim);
   timestampBarrier("Initialization Finished\n");
   timestampBarrier("Starting simulation\n");

   const int nSteps = sim->nSteps, printRate = sim->printRate;
   int iStep = 0;
   profileStart(loopTimer);
   for (; iStep < nSteps;) {
      startTimer(commReduceTimer);
      sumAtoms(sim);
      stopTimer(commReduceTimer);
      printThings(sim, iStep, getElapsedTime(timestepTimer));
      startTimer(timestepTimer);
      timestep(sim, printRate, sim->dt);
      stopTimer(timestepTimer);
      iStep += printRate;
   }
   profileStop(loopTimer);
   sumAtoms(sim);
   printThings(sim, iStep, getElapsedTime(timestepTimer));
   timestampBarrier("Ending simulation\n");

   validateResult(validate, sim);
   profileStop(totalTimer);
   printPerformanceResults(sim->atoms->nGlobal, sim->printRate);
   printPerformanceResultsYaml(yamlFile);
   destroySimulation(&sim);
   free(validate);
   yamlEnd();
   timestampBarrier("CoMD Ending\n");
   return 0;
}

/* ---- embedding API -------------------------------------------------------------------------------------------- */
SimFlat* comdCreate(int argc, char** argv)
{
   Command cmd = parseCommandLine(argc, argv);
   SimFlat* sim = initSimulation(cmd);
   if (cmd.deviceTimers || (getenv("COMD_DEVICE_TIMERS") && atoi(getenv("COMD_DEVICE_TIMERS")) != 0))
      timersUseDevice(1, sim->gpu.boundary_stream, cmd.doeam ? 176.0 * sizeof(real_t) / 8.0 : 56.0 * sizeof(real_t) / 8.0);      /* force bytes per atom */
   sumAtoms(sim);
   return sim;
}

SimFlat* comdCreateHostOnly(int argc, char** argv)
{
   Command cmd = parseCommandLine(argc, argv);
   return initSimulationHost(cmd);
}

const HostAtoms* comdHostAtoms(SimFlat* s) { return &s->atoms->h; }

int comdNeighborListBuilds(SimFlat* s) { return s->nlBuilds; }

/* EAM table `which` (0 phi, 1 rho, 2 F) as the device receives it: n, x0, invDx and the n + 3 padded samples (values[0] = leading pad) */
int comdEamTable(SimFlat* s, int which, double* x0, double* invDx, double* values)
{
   if (!s->cmdDoeam) return 0;
   EamPotential* e = (EamPotential*)s->pot;
   InterpolationObject* t = which == 0 ? e->phi : which == 1 ? e->rho : e->f;
   *x0 = t->x0; *invDx = t->invDx;
   if (values) for (int i = -1; i <= t->n + 1; ++i) values[i + 1] = t->values[i];
   return t->n;
}

void comdGridInfo(SimFlat* s, int out[6])
{
   for (int a = 0; a < 3; ++a) out[a] = s->boxes->gridSize[a];
   out[3] = s->boxes->nLocalBoxes; out[4] = s->boxes->nTotalBoxes; out[5] = s->boxes->maxAtoms;
}

int comdSimBoxFromTuple(SimFlat* s, int ix, int iy, int iz) { return getBoxFromTuple(s->boxes, ix, iy, iz); }
int comdSimBoxFromCoord(SimFlat* s, const double r[3]) { const real_t rr[3] = { (real_t)r[0], (real_t)r[1], (real_t)r[2] }; return getBoxFromCoord(s->boxes, rr); }

/* kind 0 = atom exchange list, 1 = force send list, 2 = force receive list; list may be NULL to query the size */
int comdFaceCells(SimFlat* s, int kind, int face, int* list)
{
   const int* g = s->boxes->gridSize;
   int n, *cells;
   if (kind == 0) {
      const int sz[3] = { (g[1] + 2) * (g[2] + 2), (g[0] + 2) * (g[2] + 2), (g[0] + 2) * (g[1] + 2) };
      n = 2 * sz[face / 2];
      cells = mkAtomCellList(s->boxes, (enum HaloFaceOrder)face, n);
   } else {
      const int sz[3] = { g[1] * g[2], (g[0] + 2) * g[2], (g[0] + 2) * (g[1] + 2) };
      n = sz[face / 2];
      cells = kind == 1 ? mkForceSendCellList(s->boxes, face, n) : mkForceRecvCellList(s->boxes, face, n);
   }
   if (list) memcpy(list, cells, (size_t)n * sizeof(int));
   free(cells);
   return n;
}

/* neighbour ranks of the six faces (haloExchange.c:1368-1390) and this rank's grid coordinates */
void comdNeighborRanks(SimFlat* s, int nbr[6], int coord[3])
{
   nbr[0] = processorNum(s->domain, -1, 0, 0); nbr[1] = processorNum(s->domain, +1, 0, 0);
   nbr[2] = processorNum(s->domain, 0, -1, 0); nbr[3] = processorNum(s->domain, 0, +1, 0);
   nbr[4] = processorNum(s->domain, 0, 0, -1); nbr[5] = processorNum(s->domain, 0, 0, +1);
   for (int a = 0; a < 3; ++a) coord[a] = s->domain->procCoord[a];
}

/* Drive the atom halo exchange (exchangeData x 3 axes) over HOST buffers with caller-supplied loadBuffer/unloadBuffer.
 * The production plugins pack/unpack on the device; this entry lets a CPU-only program exercise the routing, ordering
 * and transport logic with its own pack/unpack (the multi-process CPU tests do, over torch.distributed/gloo). */
void comdHaloExchangeHost(SimFlat* s, int (*load)(void*, void*, int, char*), void (*unload)(void*, void*, int, int, char*))
{
   HaloExchange* hh = initAtomHaloExchange(s->domain, s->boxes, 0);
   hh->bufCapacity = ((AtomExchangeParms*)hh->parms)->capacityAtoms * 64 + 64;
   hh->sendBufM = (char*)malloc((size_t)hh->bufCapacity); hh->sendBufP = (char*)malloc((size_t)hh->bufCapacity);
   hh->recvBufM = (char*)malloc((size_t)hh->bufCapacity); hh->recvBufP = (char*)malloc((size_t)hh->bufCapacity);
   hh->loadBuffer = load; hh->unloadBuffer = unload;
   haloExchange(hh, s);
   free(hh->sendBufM); free(hh->sendBufP); free(hh->recvBufM); free(hh->recvBufP);
   destroyHaloExchange(&hh);
}

void comdFaceShift(SimFlat* s, int face, double out[3])
{
   HaloExchange* hh = initAtomHaloExchange(s->domain, s->boxes, 0);
   for (int a = 0; a < 3; ++a) out[a] = ((AtomExchangeParms*)hh->parms)->shift[face][a];
   destroyHaloExchange(&hh);
}

int comdPutAtomInBox(SimFlat* s, int gid, int type, const double r[3], const double p[3])
{
   return putAtomInBox(s->boxes, s->atoms, gid, type, r[0], r[1], r[2], p[0], p[1], p[2]);
}

void comdDestroy(SimFlat* s) { destroySimulation(&s); }
SimGpu* comdSimGpu(SimFlat* s) { return &s->gpu; }

void comdGetEnergy(SimFlat* s, double out[3]) { out[0] = s->ePotential; out[1] = s->eKinetic; out[2] = (double)s->atoms->nGlobal; }
int comdNumGlobal(SimFlat* s) { return s->atoms->nGlobal; }
int comdNumLocalSlots(SimFlat* s) { return s->boxes->nTotalBoxes * s->boxes->maxAtoms; }

const HostAtoms* comdFetchAtoms(SimFlat* s)
{
   GetDataFromGpu(&s->gpu, &s->atoms->h);
   return &s->atoms->h;
}

void comdGatherByGid(SimFlat* s, int which, double* out)
{
   const HostAtoms* h = comdFetchAtoms(s);
   const int cap = s->boxes->maxAtoms;
   real_t* extra = NULL;
   if (which >= 4) {
      if (!s->gpu.do_eam) return;
      extra = (real_t*)malloc((size_t)comdNumLocalSlots(s) * sizeof(real_t));
      comdMemcpyDtoH(extra, which == 4 ? s->gpu.eam_pot.rhobar : s->gpu.eam_pot.dfEmbed, (long)comdNumLocalSlots(s) * sizeof(real_t));
   }
   for (int b = 0; b < s->boxes->nLocalBoxes; ++b)
      for (int o = b * cap, e = o + h->nAtoms[b]; o < e; ++o) {
         const int g = h->gid[o];
         switch (which) {
            case 0: out[3*g] = h->rx[o]; out[3*g+1] = h->ry[o]; out[3*g+2] = h->rz[o]; break;
            case 1: out[3*g] = h->px[o]; out[3*g+1] = h->py[o]; out[3*g+2] = h->pz[o]; break;
            case 2: out[3*g] = h->fx[o]; out[3*g+1] = h->fy[o]; out[3*g+2] = h->fz[o]; break;
            case 3: out[g] = h->e[o]; break;
            default: out[g] = extra[o]; break;
         }
      }
   free(extra);
}

void comdScatterByGid(SimFlat* s, int which, const double* in)
{
   HostAtoms* h = (HostAtoms*)comdFetchAtoms(s);
   const int cap = s->boxes->maxAtoms;
   for (int b = 0; b < s->boxes->nLocalBoxes; ++b)
      for (int o = b * cap, e = o + h->nAtoms[b]; o < e; ++o) {
         const int g = h->gid[o];
         if (which == 0) { h->rx[o] = in[3*g]; h->ry[o] = in[3*g+1]; h->rz[o] = in[3*g+2]; }
         else            { h->px[o] = in[3*g]; h->py[o] = in[3*g+1]; h->pz[o] = in[3*g+2]; }
      }
   CopyDataToGpu(&s->gpu, h);
}
