/* cli.c -- the CoMD command line (mycommand.c:180-320 + cmdLineParser.c): same long names, same single-letter
 * flags, same defaults, same "Command Line Parameters" YAML block.  Table-driven over getopt_long.
 * Extensions (long options only): --maxAtoms N (link-cell slot capacity; the reference fixes it at
 * compile time with -DMAXATOMS), --maxNeighbors N (Verlet-list rows per atom for the *_nl methods; the reference's
 * MAXNEIGHBORLISTSIZE), --quiet, --ljCutoffSigmas F (the reference hard-wires 5; 2.5 meets its documented LJ cohesive energy),
 * --deviceTimers (also COMD_DEVICE_TIMERS=1: HIP-event timing of the phases in the reference's timer table). */
#include "comd_host.h"
#include <getopt.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const char* longName; char shortName; int hasArg; char type; void* ptr; int size; const char* help; } ArgDef;

static void printArgs(const ArgDef* defs, int n)
{
   fprintf(screenOut, "\n  Arguments are: \n");
   for (int i = 0; i < n; ++i) {
      char shortBuf[8] = "  ";
      if (defs[i].shortName > 0) snprintf(shortBuf, sizeof shortBuf, "-%c", defs[i].shortName);
      fprintf(screenOut, "   --%-20s %s  arg=%1d type=%c  %s\n", defs[i].longName, shortBuf, defs[i].hasArg, defs[i].type, defs[i].help);
   }
   fprintf(screenOut, "\n\n");
}

Command parseCommandLine(int argc, char** argv)
{
   Command cmd;
   memset(&cmd, 0, sizeof cmd);
   strcpy(cmd.potDir, "pots");
   strcpy(cmd.potType, "funcfl");
   strcpy(cmd.method, "thread_atom");
   cmd.nx = cmd.ny = cmd.nz = 20;
   cmd.xproc = cmd.yproc = cmd.zproc = 1;
   cmd.nSteps = 100; cmd.printRate = 10;
   cmd.ljCutoffSigmas = 5.0;
   cmd.dt = 1.0; cmd.lat = -1.0; cmd.temperature = 600.0; cmd.initialDelta = 0.0; cmd.relativeSkinDistance = 0.1;
   int help = 0;

   const ArgDef defs[] = {
      { "help",         'h', 0, 'i', &help,               0, "print this message" },
      { "potDir",       'd', 1, 's', cmd.potDir,  sizeof cmd.potDir,  "potential directory" },
      { "potName",      'p', 1, 's', cmd.potName, sizeof cmd.potName, "potential name" },
      { "potType",      't', 1, 's', cmd.potType, sizeof cmd.potType, "potential type (funcfl or setfl)" },
      { "doeam",        'e', 0, 'i', &cmd.doeam,          0, "compute eam potentials" },
      { "nx",           'x', 1, 'i', &cmd.nx,             0, "number of unit cells in x" },
      { "ny",           'y', 1, 'i', &cmd.ny,             0, "number of unit cells in y" },
      { "nz",           'z', 1, 'i', &cmd.nz,             0, "number of unit cells in z" },
      { "xproc",        'i', 1, 'i', &cmd.xproc,          0, "processors in x direction" },
      { "yproc",        'j', 1, 'i', &cmd.yproc,          0, "processors in y direction" },
      { "zproc",        'k', 1, 'i', &cmd.zproc,          0, "processors in z direction" },
      { "nSteps",       'N', 1, 'i', &cmd.nSteps,         0, "number of time steps" },
      { "printRate",    'n', 1, 'i', &cmd.printRate,      0, "number of steps between output" },
      { "dt",           'D', 1, 'd', &cmd.dt,             0, "time step (in fs)" },
      { "lat",          'l', 1, 'd', &cmd.lat,            0, "lattice parameter (Angstroms)" },
      { "temp",         'T', 1, 'd', &cmd.temperature,    0, "initial temperature (K)" },
      { "delta",        'r', 1, 'd', &cmd.initialDelta,   0, "initial delta (Angstroms)" },
      { "hilbert",      'H', 0, 'i', &cmd.doHilbert,      0, "space-filling (Hilbert) curve for the numbering of the link cells" },
      { "skinDistance", 'S', 1, 'd', &cmd.relativeSkinDistance, 0, "skinDistance (relative to cutoff (default: 0.1))" },
      { "method",       'm', 1, 's', cmd.method,  sizeof cmd.method,  "thread_atom,thread_atom_nl,cta_cell (warp_atom[_nl] run as thread_atom[_nl])" },
      { "gpuAsync",     'a', 1, 'i', &cmd.gpuAsync,       0, "communicaton hiding optimization using streams" },
      { "gpuProfile",   's', 0, 'i', &cmd.gpuProfile,     0, "profiling mode: reboxing disabled, single kernel run" },
      { "ljInterpolation", 'I', 0, 'i', &cmd.ljInterpolation, 0, "Lennard-Jones by table interpolation (not supported)" },
      { "spline",       'P', 0, 'i', &cmd.spline,         0, "cubic-spline EAM tables in r^2 (thread_atom, cta_cell)" },
      { "usePairlist",  'L', 0, 'i', &cmd.usePairlist,    0, "pairlists for cta_cell LJ" },
      { "maxAtoms",      0,  1, 'i', &cmd.maxAtoms,       0, "link-cell slot capacity (0 = from the lattice)" },
      { "maxNeighbors",  0,  1, 'i', &cmd.maxNeighbors,   0, "neighbour-list rows per atom for *_nl (0 = from cutoff + skin)" },
      { "quiet",         0,  0, 'i', &cmd.quiet,          0, "no stdout report" },
      { "ljCutoffSigmas", 0, 1, 'd', &cmd.ljCutoffSigmas, 0, "LJ cutoff in sigmas (5 as in ljForce.c:114; 2.5 reproduces the cohesive energy of CoMD.c:897)" },
      { "deviceTimers",  0,  0, 'i', &cmd.deviceTimers,   0, "time the phases of timestep() with HIP events on the device (the host timers of the reference see launches, not kernels)" },
   };
   const int nDefs = (int)(sizeof defs / sizeof defs[0]);

   struct option* longOpts = (struct option*)calloc((size_t)nDefs + 1, sizeof(struct option));
   char shortOpts[4 * 32]; int so = 0;
   for (int i = 0; i < nDefs; ++i) {
      longOpts[i].name = defs[i].longName;
      longOpts[i].has_arg = defs[i].hasArg ? required_argument : no_argument;
      longOpts[i].flag = NULL;
      longOpts[i].val = defs[i].shortName ? defs[i].shortName : 1000 + i;
      if (defs[i].shortName) { shortOpts[so++] = defs[i].shortName; if (defs[i].hasArg) shortOpts[so++] = ':'; }
   }
   shortOpts[so] = '\0';

   optind = 1;            /* allow repeated parsing inside one process (library use) */
   int c, idx;
   while ((c = getopt_long(argc, argv, shortOpts, longOpts, &idx)) != -1) {
      const ArgDef* d = NULL;
      for (int i = 0; i < nDefs; ++i) if (longOpts[i].val == c) { d = &defs[i]; break; }
      if (!d) { fprintf(screenOut, "\n\n    invalid switch : -%c in getopt()\n\n\n", optopt); continue; }
      if (!d->hasArg) { *(int*)d->ptr = 1; continue; }
      switch (d->type) {
         case 'i': *(int*)d->ptr = atoi(optarg); break;
         case 'd': *(double*)d->ptr = atof(optarg); break;
         case 's': strncpy((char*)d->ptr, optarg, (size_t)d->size - 1); ((char*)d->ptr)[d->size - 1] = '\0'; break;
      }
   }
   free(longOpts);

   if (strlen(cmd.potName) == 0) {
      if (strcmp(cmd.potType, "setfl") == 0) strcpy(cmd.potName, "Cu01.eam.alloy");
      if (strcmp(cmd.potType, "funcfl") == 0) strcpy(cmd.potName, "Cu_u6.eam");
   }
   if (help) { printArgs(defs, nDefs); exit(2); }
   return cmd;
}

void printCmdYaml(FILE* file, Command* cmd)
{
   if (!printRank() || !file) return;
   fprintf(file,
           "Command Line Parameters:\n"
           "  doeam: %d\n"
           "  potDir: %s\n"
           "  potName: %s\n"
           "  potType: %s\n"
           "  nx: %d\n"
           "  ny: %d\n"
           "  nz: %d\n"
           "  xproc: %d\n"
           "  yproc: %d\n"
           "  zproc: %d\n"
           "  Lattice constant: %g Angstroms\n"
           "  nSteps: %d\n"
           "  printRate: %d\n"
           "  Time step: %g fs\n"
           "  Initial Temperature: %g K\n"
           "  Initial Delta: %g Angstroms\n\n"
           "  GPU async opt: %d\n"
           "  GPU profiling mode: %d\n"
           "  GPU method: %s\n"
           "  Space-filling (Hilbert): %d\n"
           "\n",
           cmd->doeam, cmd->potDir, cmd->potName, cmd->potType, cmd->nx, cmd->ny, cmd->nz,
           cmd->xproc, cmd->yproc, cmd->zproc, cmd->lat, cmd->nSteps, cmd->printRate, cmd->dt,
           cmd->temperature, cmd->initialDelta, cmd->gpuAsync, cmd->gpuProfile, cmd->method, cmd->doHilbert);
   fflush(file);
}
