// comm_rccl.hip -- RCCL-over-xGMI transport behind CommTransport (include/comd_hip.h).
//
// Takes the place of the reference's comm.cc (libmp / GPUDirect-Async wrappers, :326-657) and of MPI_Sendrecv /
// MPI_Allreduce in parallel.c:100-193.  One process per GPU; halo messages are device buffers exchanged with
// ncclSend/ncclRecv pairs on the caller's stream.  Message sizes are data dependent and RCCL has no probe, so each
// paired exchange first swaps the two byte counts (one int each way) and then the payloads -- or, when both ends already agree
// on the sizes (sendrecv2sized: the halo driver's speculative protocol), posts the payloads alone with no host synchronisation.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/stat.h>
#include <string>

#include "comd_hip.h"

#define HIPC(cmd) do { hipError_t e_ = (cmd); if (e_ != hipSuccess) { fprintf(stderr, "Rank %d: HIP error %s at %s:%d\n", g_rank, hipGetErrorString(e_), __FILE__, __LINE__); exit(-1); } } while (0)
#define NCCLC(cmd) do { ncclResult_t r_ = (cmd); if (r_ != ncclSuccess) { fprintf(stderr, "Rank %d: RCCL error %s at %s:%d\n", g_rank, ncclGetErrorString(r_), __FILE__, __LINE__); exit(-1); } } while (0)

static ncclComm_t g_comm = nullptr;
static int g_rank = 0, g_nRanks = 1;
static hipStream_t g_stream = nullptr;      // reductions / broadcasts
static int* g_dSizes = nullptr;             // device [4]: two send sizes, two receive sizes
static int* g_hSizes = nullptr;             // pinned [4]
static void* g_dScratch = nullptr;          // device scratch for small host-buffer collectives
static const size_t kScratchBytes = 1 << 20;
static std::string g_idFile;

static int rcclSendrecv(void*, const void* sendBuf, int sendLen, int dest, void* recvBuf, int recvCap, int source, int device, comdStream_t stream)
{
   hipStream_t st = (hipStream_t)stream;
   if (!device) { fprintf(stderr, "Rank %d: RCCL transport moves device buffers only\n", g_rank); exit(-1); }
   g_hSizes[0] = sendLen;
   HIPC(hipMemcpyAsync(g_dSizes, g_hSizes, sizeof(int), hipMemcpyHostToDevice, st));
   NCCLC(ncclGroupStart());
   NCCLC(ncclSend(g_dSizes, 1, ncclInt, dest, g_comm, st));
   NCCLC(ncclRecv(g_dSizes + 1, 1, ncclInt, source, g_comm, st));
   NCCLC(ncclGroupEnd());
   HIPC(hipMemcpyAsync(g_hSizes + 1, g_dSizes + 1, sizeof(int), hipMemcpyDeviceToHost, st));
   HIPC(hipStreamSynchronize(st));
   const int recvLen = g_hSizes[1];
   if (recvLen > recvCap) { fprintf(stderr, "Rank %d: incoming halo message (%d B) exceeds the buffer (%d B)\n", g_rank, recvLen, recvCap); exit(-1); }
   NCCLC(ncclGroupStart());
   if (sendLen > 0) NCCLC(ncclSend(sendBuf, (size_t)sendLen, ncclChar, dest, g_comm, st));
   if (recvLen > 0) NCCLC(ncclRecv(recvBuf, (size_t)recvLen, ncclChar, source, g_comm, st));
   NCCLC(ncclGroupEnd());
   return recvLen;
}

// both faces of an axis phase: one size handshake (two ints each way), one payload group of up to four transfers
static void rcclSendrecv2(void*, const void* sendM, int nSendM, int dstM, void* recvP, const void* sendP, int nSendP, int dstP, void* recvM,
                          int recvCap, int device, comdStream_t stream, int nRecv[2])
{
   hipStream_t st = (hipStream_t)stream;
   if (!device) { fprintf(stderr, "Rank %d: RCCL transport moves device buffers only\n", g_rank); exit(-1); }
   g_hSizes[0] = nSendM; g_hSizes[1] = nSendP;
   HIPC(hipMemcpyAsync(g_dSizes, g_hSizes, 2 * sizeof(int), hipMemcpyHostToDevice, st));
   NCCLC(ncclGroupStart());
   NCCLC(ncclSend(g_dSizes, 1, ncclInt, dstM, g_comm, st));          // my minus-face size -> minus neighbour
   NCCLC(ncclSend(g_dSizes + 1, 1, ncclInt, dstP, g_comm, st));      // my plus-face size  -> plus neighbour
   NCCLC(ncclRecv(g_dSizes + 2, 1, ncclInt, dstP, g_comm, st));      // plus neighbour's minus-face size  (arrives in recvP)
   NCCLC(ncclRecv(g_dSizes + 3, 1, ncclInt, dstM, g_comm, st));      // minus neighbour's plus-face size  (arrives in recvM)
   NCCLC(ncclGroupEnd());
   HIPC(hipMemcpyAsync(g_hSizes + 2, g_dSizes + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
   HIPC(hipStreamSynchronize(st));
   const int nP = g_hSizes[2], nM = g_hSizes[3];
   if (nP > recvCap || nM > recvCap) { fprintf(stderr, "Rank %d: incoming halo message (%d / %d B) exceeds the buffer (%d B)\n", g_rank, nP, nM, recvCap); exit(-1); }
   NCCLC(ncclGroupStart());
   if (nSendM > 0) NCCLC(ncclSend(sendM, (size_t)nSendM, ncclChar, dstM, g_comm, st));
   if (nSendP > 0) NCCLC(ncclSend(sendP, (size_t)nSendP, ncclChar, dstP, g_comm, st));
   if (nP > 0) NCCLC(ncclRecv(recvP, (size_t)nP, ncclChar, dstP, g_comm, st));
   if (nM > 0) NCCLC(ncclRecv(recvM, (size_t)nM, ncclChar, dstM, g_comm, st));
   NCCLC(ncclGroupEnd());
   nRecv[0] = nP; nRecv[1] = nM;
}

// Both ends of every message know its size: the four transfers are enqueued and the host moves on.  With dstM == dstP (a 2-rank axis,
// e.g. every axis of a 2x2x2 grid) the peer posts its sends in the same order, so its minus-face message meets the first receive
// (recvP) and its plus-face message the second (recvM): RCCL matches sends and receives between two ranks in posting order.
static void rcclSendrecv2Sized(void*, const void* sendM, int nSendM, int dstM, void* recvP, int nRecvP, const void* sendP, int nSendP, int dstP,
                               void* recvM, int nRecvM, int device, comdStream_t stream)
{
   hipStream_t st = (hipStream_t)stream;
   if (!device) { fprintf(stderr, "Rank %d: RCCL transport moves device buffers only\n", g_rank); exit(-1); }
   NCCLC(ncclGroupStart());
   if (nSendM > 0) NCCLC(ncclSend(sendM, (size_t)nSendM, ncclChar, dstM, g_comm, st));
   if (nSendP > 0) NCCLC(ncclSend(sendP, (size_t)nSendP, ncclChar, dstP, g_comm, st));
   if (nRecvP > 0) NCCLC(ncclRecv(recvP, (size_t)nRecvP, ncclChar, dstP, g_comm, st));
   if (nRecvM > 0) NCCLC(ncclRecv(recvM, (size_t)nRecvM, ncclChar, dstM, g_comm, st));
   NCCLC(ncclGroupEnd());
}

static void rcclAllreduce(void*, void* buf, int count, int dtype)
{
   const size_t bytes = (size_t)count * (dtype == 1 ? sizeof(double) : dtype == 3 ? sizeof(float) : sizeof(int));
   if (bytes > kScratchBytes) { fprintf(stderr, "Rank %d: allreduce of %zu bytes exceeds the scratch buffer\n", g_rank, bytes); exit(-1); }
   HIPC(hipMemcpyAsync(g_dScratch, buf, bytes, hipMemcpyHostToDevice, g_stream));
   NCCLC(ncclAllReduce(g_dScratch, g_dScratch, (size_t)count, dtype == 1 ? ncclDouble : dtype == 3 ? ncclFloat : ncclInt, dtype == 2 ? ncclMax : ncclSum, g_comm, g_stream));
   HIPC(hipMemcpyAsync(buf, g_dScratch, bytes, hipMemcpyDeviceToHost, g_stream));
   HIPC(hipStreamSynchronize(g_stream));
}

static void rcclBcast(void*, void* buf, int len, int root)
{
   if ((size_t)len > kScratchBytes) { fprintf(stderr, "Rank %d: bcast of %d bytes exceeds the scratch buffer\n", g_rank, len); exit(-1); }
   HIPC(hipMemcpyAsync(g_dScratch, buf, (size_t)len, hipMemcpyHostToDevice, g_stream));
   NCCLC(ncclBroadcast(g_dScratch, g_dScratch, (size_t)len, ncclChar, root, g_comm, g_stream));
   HIPC(hipMemcpyAsync(buf, g_dScratch, (size_t)len, hipMemcpyDeviceToHost, g_stream));
   HIPC(hipStreamSynchronize(g_stream));
}

static void rcclBarrier(void*)
{
   int one = 1;
   rcclAllreduce(nullptr, &one, 1, 0);
}

extern "C" int comdCommGetUniqueId(char* id128)
{
   static_assert(sizeof(ncclUniqueId) <= COMD_UNIQUE_ID_BYTES, "unique id larger than the ABI slot");
   ncclUniqueId id;
   if (ncclGetUniqueId(&id) != ncclSuccess) return -1;
   memset(id128, 0, COMD_UNIQUE_ID_BYTES);
   memcpy(id128, &id, sizeof id);
   return 0;
}

extern "C" int comdCommInitRank(const char* id128, int rank, int nRanks, CommTransport* out)
{
   g_rank = rank; g_nRanks = nRanks;
   ncclUniqueId id;
   memcpy(&id, id128, sizeof id);
   {  // a communicator that cannot be formed is reported, not fatal: the caller may fall back to another transport
      ncclResult_t r = ncclCommInitRank(&g_comm, nRanks, id, rank);
      if (r != ncclSuccess) { fprintf(stderr, "Rank %d: ncclCommInitRank failed: %s\n", rank, ncclGetErrorString(r)); g_comm = nullptr; return -1; }
   }
   HIPC(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
   HIPC(hipMalloc((void**)&g_dSizes, 4 * sizeof(int)));
   HIPC(hipHostMalloc((void**)&g_hSizes, 4 * sizeof(int), hipHostMallocDefault));
   HIPC(hipMalloc(&g_dScratch, kScratchBytes));
   out->ctx = nullptr;
   out->sendrecv = rcclSendrecv;
   out->sendrecv2 = rcclSendrecv2;
   out->sendrecv2sized = rcclSendrecv2Sized;
   out->allreduce = rcclAllreduce;
   out->bcast = rcclBcast;
   out->barrier = rcclBarrier;
   return 0;
}

extern "C" int comdCommInitFromEnv(CommTransport* out, int* rank, int* nRanks, int* localRank)
{
   const char* r = getenv("RANK"); const char* w = getenv("WORLD_SIZE"); const char* l = getenv("LOCAL_RANK");
   if (!r || !w) return -1;
   *rank = atoi(r); *nRanks = atoi(w); *localRank = l ? atoi(l) : *rank;
   const char* dir = getenv("COMD_RDZV_DIR"); const char* port = getenv("MASTER_PORT"); const char* run = getenv("TORCHELASTIC_RUN_ID");
   // One file per LAUNCH: the ranks of a launch share their launcher (torchrun's agent, mpirun, a shell) as parent process, a relaunch
   // or a later run has another one, and torchrun counts its restarts -- a file left behind by a crashed run is never picked up.
   // COMD_RDZV_TOKEN overrides the parent pid for launchers that give every rank its own parent.
   const char* token = getenv("COMD_RDZV_TOKEN"); const char* restart = getenv("TORCHELASTIC_RESTART_COUNT");
   g_idFile = std::string(dir ? dir : "/tmp") + "/comd_rccl_id_" + (port ? port : "29500") + "_" + (run ? run : "0") + "_" + (restart ? restart : "0") +
              "_" + (token ? std::string(token) : std::to_string((long)getppid()));
   char id[COMD_UNIQUE_ID_BYTES];
   if (*rank == 0) {
      unlink(g_idFile.c_str());
      if (comdCommGetUniqueId(id) != 0) return -1;
      std::string tmp = g_idFile + ".tmp";
      FILE* f = fopen(tmp.c_str(), "wb");
      if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return -1;
      fclose(f);
      if (rename(tmp.c_str(), g_idFile.c_str()) != 0) return -1;
   } else {
      const time_t start = time(nullptr);
      for (int tries = 0; ; ++tries) {
         struct stat sb;
         if (stat(g_idFile.c_str(), &sb) == 0 && sb.st_size == (off_t)sizeof id && sb.st_mtime + 120 >= start) {
            FILE* f = fopen(g_idFile.c_str(), "rb");
            if (f && fread(id, 1, sizeof id, f) == sizeof id) { fclose(f); break; }
            if (f) fclose(f);
         }
         if (tries > 1200) { fprintf(stderr, "Rank %d: timed out waiting for %s\n", *rank, g_idFile.c_str()); return -1; }
         usleep(100000);
      }
   }
   return comdCommInitRank(id, *rank, *nRanks, out);
}

extern "C" int comdCommInfo(int* nRanks, int* rank, int* device)
{
   if (!g_comm) return -1;
   int n = 0, r = 0, d = 0;
   if (ncclCommCount(g_comm, &n) != ncclSuccess || ncclCommUserRank(g_comm, &r) != ncclSuccess || ncclCommCuDevice(g_comm, &d) != ncclSuccess) return -1;
   if (nRanks) *nRanks = n;
   if (rank) *rank = r;
   if (device) *device = d;
   return 0;
}

extern "C" void comdCommFinalize(void)
{
   if (!g_comm) return;
   rcclBarrier(nullptr);
   if (g_rank == 0 && !g_idFile.empty()) unlink(g_idFile.c_str());
   ncclCommDestroy(g_comm); g_comm = nullptr;
   if (g_dSizes) (void)hipFree(g_dSizes);
   if (g_hSizes) (void)hipHostFree(g_hSizes);
   if (g_dScratch) (void)hipFree(g_dScratch);
   if (g_stream) (void)hipStreamDestroy(g_stream);
   g_dSizes = nullptr; g_hSizes = nullptr; g_dScratch = nullptr; g_stream = nullptr;
}
