/* timers.c -- the reference's twelve wall-clock timers and its end-of-run report
 * (performanceTimers.c:55-340: names, per-rank table, cross-rank statistics, three rate figures,
 * YAML block).  Like the reference's, these are host timers: GPU work lands in whichever timer
 * hits the next blocking call (performanceTimers.h:29-44).  With --deviceTimers (timersUseDevice) every
 * start/stop of a phase inside timestep() also records a HIP event on the simulation's stream, and the
 * report shows DEVICE seconds for those phases -- same table, same YAML keys -- plus two keys the
 * reference lacks: atomUpdatesPerSec and forceKernelGBs. */
#include "comd_host.h"
#include <inttypes.h>
#include <math.h>
#include <string.h>
#include <sys/time.h>
#include <stdlib.h>

static const char* timerName[numberOfTimers] = {
   "total", "loop", "timestep", "  position", "  velocity", "  redistribute", "    atomHalo",
   "  force", "    eamHalo", "commHalo", "commReduce", "  neighborList"
};

typedef struct { uint64_t start, total, count, elapsed; int minRank, maxRank; double minValue, maxValue, average, stdev; } Timers;
static Timers perfTimer[numberOfTimers];
static struct { double atomRate, atomAllRate, atomsPerUSec; } perfGlobal;

static uint64_t getTime(void)
{
   struct timeval t;
   gettimeofday(&t, NULL);
   return UINT64_C(1000000) * (uint64_t)t.tv_sec + (uint64_t)t.tv_usec;
}
static double getTick(void) { return 1.0e-6; }

/* ---- device timing: an event pair per start/stop, read back in batches ---- */
typedef struct { void* a; void* b; int h; } DevPair;
static int devOn = 0;
static comdStream_t devStream = NULL;
static double devForceBytes = 0.0;
static DevPair* devPairs = NULL; static int nDev = 0, capDev = 0;
static void* devOpen[numberOfTimers];
static double devMs[numberOfTimers];
static void** devFree = NULL; static int nFree = 0, capFree = 0;

static int devicePhase(enum TimerHandle h)
{
   return h == positionTimer || h == velocityTimer || h == redistributeTimer || h == atomHaloTimer || h == computeForceTimer || h == eamHaloTimer
          || h == neighborListBuildTimer;
}
static void* devEvent(void) { return nFree > 0 ? devFree[--nFree] : comdEventCreate(); }
static void devRelease(void* e)
{
   if (nFree == capFree) { capFree = capFree ? 2 * capFree : 256; devFree = (void**)realloc(devFree, (size_t)capFree * sizeof(void*)); }
   devFree[nFree++] = e;
}
static void devFlush(void)
{
   for (int i = 0; i < nDev; ++i) {
      devMs[devPairs[i].h] += comdEventElapsedMs(devPairs[i].a, devPairs[i].b);      /* waits for the stop event */
      devRelease(devPairs[i].a); devRelease(devPairs[i].b);
   }
   nDev = 0;
}
void timersUseDevice(int on, comdStream_t stream, double forceBytesPerAtom)
{
   devOn = on; devStream = stream; devForceBytes = forceBytesPerAtom;
   memset(devMs, 0, sizeof devMs); memset(devOpen, 0, sizeof devOpen);
}

void resetTimers(void) { memset(perfTimer, 0, sizeof perfTimer); }
void profileStart(enum TimerHandle h)
{
   perfTimer[h].start = getTime();
   if (devOn && devicePhase(h)) { devOpen[h] = devEvent(); comdEventRecord(devOpen[h], devStream); }
}
void profileStop(enum TimerHandle h)
{
   uint64_t delta = getTime() - perfTimer[h].start;
   perfTimer[h].count += 1; perfTimer[h].total += delta; perfTimer[h].elapsed += delta;
   if (devOn && devicePhase(h) && devOpen[h]) {
      void* b = devEvent();
      comdEventRecord(b, devStream);
      if (nDev == capDev) { capDev = capDev ? 2 * capDev : 1024; devPairs = (DevPair*)realloc(devPairs, (size_t)capDev * sizeof(DevPair)); }
      devPairs[nDev].a = devOpen[h]; devPairs[nDev].b = b; devPairs[nDev].h = h; nDev++;
      devOpen[h] = NULL;
      if (nDev >= 4096) devFlush();
   }
}
double getElapsedTime(enum TimerHandle h)
{
   double t = getTick() * (double)perfTimer[h].elapsed;
   perfTimer[h].elapsed = 0;
   return t;
}

/* min/max with owning rank: done with two sum-reductions over a one-hot layout (no MINLOC in the transport) */
static void timerStats(void)
{
   const int n = getNRanks(), me = getMyRank();
   double send[numberOfTimers], recv[numberOfTimers];
   for (int i = 0; i < numberOfTimers; ++i) send[i] = (double)perfTimer[i].total;
   addDoubleParallel(send, recv, numberOfTimers);
   for (int i = 0; i < numberOfTimers; ++i) perfTimer[i].average = recv[i] / n;
   double* all = (double*)calloc((size_t)n * numberOfTimers, sizeof(double));
   double* allSum = (double*)calloc((size_t)n * numberOfTimers, sizeof(double));
   for (int i = 0; i < numberOfTimers; ++i) all[(size_t)me * numberOfTimers + i] = (double)perfTimer[i].total;
   addDoubleParallel(all, allSum, n * numberOfTimers);
   for (int i = 0; i < numberOfTimers; ++i) {
      perfTimer[i].minValue = perfTimer[i].maxValue = allSum[i]; perfTimer[i].minRank = perfTimer[i].maxRank = 0;
      double var = 0.0;
      for (int r = 0; r < n; ++r) {
         double v = allSum[(size_t)r * numberOfTimers + i];
         if (v < perfTimer[i].minValue) { perfTimer[i].minValue = v; perfTimer[i].minRank = r; }
         if (v > perfTimer[i].maxValue) { perfTimer[i].maxValue = v; perfTimer[i].maxRank = r; }
         var += (v - perfTimer[i].average) * (v - perfTimer[i].average);
      }
      perfTimer[i].stdev = sqrt(var / n);
   }
   free(all); free(allSum);
}

static double perfAtomUpdatesPerSec = 0.0, perfForceKernelGBs = 0.0;

void printPerformanceResults(int nGlobalAtoms, int printRate)
{
   if (devOn) {      /* device seconds take the place of the host totals of the phases inside timestep() (tick = 1 us) */
      devFlush();
      for (int i = 0; i < numberOfTimers; ++i) if (devicePhase((enum TimerHandle)i)) perfTimer[i].total = (uint64_t)(devMs[i] * 1000.0 + 0.5);
   }
   timerStats();
   if (!printRank()) return;
   double tick = getTick();
   double loopTime = perfTimer[loopTimer].total * tick;
   fprintf(screenOut, "\n\nTimings for Rank %d\n", getMyRank());
   fprintf(screenOut, "        Timer        # Calls    Avg/Call (s)   Total (s)    %% Loop\n");
   fprintf(screenOut, "___________________________________________________________________\n");
   for (int i = 0; i < numberOfTimers; ++i) {
      double totalTime = perfTimer[i].total * tick;
      if (perfTimer[i].count > 0)
         fprintf(screenOut, "%-16s%12" PRIu64 "     %8.4f      %8.4f    %8.2f\n", timerName[i], perfTimer[i].count,
                 totalTime / (double)perfTimer[i].count, totalTime, totalTime / loopTime * 100.0);
   }
   fprintf(screenOut, "\nTiming Statistics Across %d Ranks:\n", getNRanks());
   fprintf(screenOut, "        Timer        Rank: Min(s)       Rank: Max(s)      Avg(s)    Stdev(s)\n");
   fprintf(screenOut, "_____________________________________________________________________________\n");
   for (int i = 0; i < numberOfTimers; ++i)
      if (perfTimer[i].count > 0)
         fprintf(screenOut, "%-16s%6d:%10.4f  %6d:%10.4f  %10.4f  %10.4f\n", timerName[i],
                 perfTimer[i].minRank, perfTimer[i].minValue * tick, perfTimer[i].maxRank, perfTimer[i].maxValue * tick,
                 perfTimer[i].average * tick, perfTimer[i].stdev * tick);
   double atomsPerTask = nGlobalAtoms / (real_t)getNRanks();
   perfGlobal.atomRate = perfTimer[timestepTimer].average * tick * 1e6 / (atomsPerTask * perfTimer[timestepTimer].count * printRate);
   perfGlobal.atomAllRate = perfTimer[timestepTimer].average * tick * 1e6 / ((double)nGlobalAtoms * perfTimer[timestepTimer].count * printRate);
   perfGlobal.atomsPerUSec = 1.0 / perfGlobal.atomAllRate;
   fprintf(screenOut, "\n---------------------------------------------------\n");
   fprintf(screenOut, " Average atom update rate:     %6.2f us/atom/task\n", perfGlobal.atomRate);
   fprintf(screenOut, "---------------------------------------------------\n\n");
   fprintf(screenOut, "\n---------------------------------------------------\n");
   fprintf(screenOut, " Average all atom update rate: %6.2f us/atom\n", perfGlobal.atomAllRate);
   fprintf(screenOut, "---------------------------------------------------\n\n");
   fprintf(screenOut, "\n---------------------------------------------------\n");
   fprintf(screenOut, " Average atom rate:            %6.2f atoms/us\n", perfGlobal.atomsPerUSec);
   fprintf(screenOut, "---------------------------------------------------\n\n");
   /* beside the reference's three figures (whose %6.2f formats were made for CPUs): the rate in atom-updates per second, and -- with device timers --
    * the algorithmic bytes of the force evaluations over the device seconds of the force phase */
   const double steps = (double)perfTimer[timestepTimer].count * printRate;
   perfAtomUpdatesPerSec = loopTime > 0.0 ? (double)nGlobalAtoms * steps / loopTime : 0.0;
   fprintf(screenOut, " Atom updates per second:      %12.4e (all ranks, loop time)\n", perfAtomUpdatesPerSec);
   if (devOn && perfTimer[computeForceTimer].total > 0) {
      perfForceKernelGBs = devForceBytes * atomsPerTask * (double)perfTimer[computeForceTimer].count / (perfTimer[computeForceTimer].total * tick) / 1e9;
      fprintf(screenOut, " Force phase, device timed:    %12.2f GB/s of algorithmic bytes (%.0f B/atom) = %.4f of the 8 TB/s HBM roof\n",
              perfForceKernelGBs, devForceBytes, perfForceKernelGBs / 8000.0);
      fprintf(screenOut, " (position, velocity, redistribute, atomHalo, force, eamHalo, neighborList above are DEVICE seconds: HIP events on the simulation's stream)\n");
   }
}

void printPerformanceResultsYaml(FILE* file)
{
   if (!printRank() || !file) return;
   double tick = getTick();
   double loopTime = perfTimer[loopTimer].total * tick;
   fprintf(file, "\nPerformance Results:\n");
   fprintf(file, "  TotalRanks: %d\n", getNRanks());
   fprintf(file, "  ReportingTimeUnits: seconds\n");
   fprintf(file, "Performance Results For Rank %d:\n", getMyRank());
   for (int i = 0; i < numberOfTimers; i++)
      if (perfTimer[i].count > 0) {
         double totalTime = perfTimer[i].total * tick;
         fprintf(file, "  Timer: %s\n", timerName[i]);
         fprintf(file, "    CallCount: %" PRIu64 "\n", perfTimer[i].count);
         fprintf(file, "    AvgPerCall: %8.4f\n", totalTime / (double)perfTimer[i].count);
         fprintf(file, "    Total:      %8.4f\n", totalTime);
         fprintf(file, "    PercentLoop: %8.2f\n", totalTime / loopTime * 100);
      }
   fprintf(file, "Performance Results Across Ranks:\n");
   for (int i = 0; i < numberOfTimers; i++)
      if (perfTimer[i].count > 0) {
         fprintf(file, "  Timer: %s\n", timerName[i]);
         fprintf(file, "    MinRank: %d\n", perfTimer[i].minRank);
         fprintf(file, "    MinTime: %8.4f\n", perfTimer[i].minValue * tick);
         fprintf(file, "    MaxRank: %d\n", perfTimer[i].maxRank);
         fprintf(file, "    MaxTime: %8.4f\n", perfTimer[i].maxValue * tick);
         fprintf(file, "    AvgTime: %8.4f\n", perfTimer[i].average * tick);
         fprintf(file, "    StdevTime: %8.4f\n", perfTimer[i].stdev * tick);
      }
   fprintf(file, "Performance Global Update Rates:\n");
   fprintf(file, "  AtomUpdateRate:\n    AverageRate: %6.2f\n    Units: us/atom/task\n", perfGlobal.atomRate);
   fprintf(file, "  AllAtomUpdateRate:\n    AverageRate: %6.2f\n    Units: us/atom\n", perfGlobal.atomAllRate);
   fprintf(file, "  AtomRate:\n    AverageRate: %6.2f\n    Units: atoms/us\n", perfGlobal.atomsPerUSec);
   fprintf(file, "  atomUpdatesPerSec: %.6e\n", perfAtomUpdatesPerSec);
   if (devOn) {
      fprintf(file, "  deviceTimers: 1\n");
      fprintf(file, "  forceKernelGBs: %.3f\n", perfForceKernelGBs);
   }
   fprintf(file, "\n");
}
