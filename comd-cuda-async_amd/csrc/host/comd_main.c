/* comd_main.c -- the `comd-hip` executable: the reference's CoMD binary (CoMD.c:86-187) for MI355X.
 * One process per GPU.  A single process needs no launcher.  Several processes are started by any launcher that
 * exports RANK, WORLD_SIZE and LOCAL_RANK (torchrun does); they bootstrap RCCL through a file under $COMD_RDZV_DIR
 * (default /tmp) keyed by MASTER_PORT -- the reference's rank->GPU binding lives in wrapper.sh / comm_select_device. */
#include "comd_host.h"
#include <stdlib.h>
#include <string.h>


int main(int argc, char** argv)
{
   CommTransport t; memset(&t, 0, sizeof t);
   int rank = 0, nRanks = 1, localRank = 0;
   const char* ws = getenv("WORLD_SIZE");
   const char* lb = getenv("COMD_LOOPBACK_TRANSPORT");
   /* WORLD_SIZE=1 with COMD_LOOPBACK_TRANSPORT=1: the one rank still bootstraps RCCL and talks to itself through it */
   int multi = ws && (atoi(ws) > 1 || (atoi(ws) == 1 && lb && atoi(lb) != 0));
   if (multi) {
      localRank = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0;
      SetupGpu(localRank, getenv("RANK") ? atoi(getenv("RANK")) : 0, 0);
      if (comdCommInitFromEnv(&t, &rank, &nRanks, &localRank) != 0) { fprintf(stderr, "RCCL bootstrap failed\n"); return 1; }
      initParallel(rank, nRanks, &t);
   } else {
      if (comdDeviceCount() < 1) { fprintf(stderr, "comd-hip: no HIP device visible (this build has no CPU path)\n"); return 1; }
      initParallel(0, 1, NULL);
      SetupGpu(0, 0, 1);
   }
   int rc = comdMain(argc, argv);
   if (multi) comdCommFinalize();
   destroyParallel();
   return rc;
}
