/* parallel.c -- rank bookkeeping and the pluggable transport behind the reference's parallel.h API
 * (parallel.c:24-193: getNRanks/getMyRank/printRank, sendReceiveParallel, addXParallel, maxIntParallel, bcastParallel,
 * barrierParallel, timestampBarrier).  The reference binds these to MPI; here one process drives one GPU and the
 * transport is a small vtable: none (single rank), RCCL over xGMI (libcomd_hip's comdComm*), or callbacks supplied
 * by the embedding program (the CPU tests plug torch.distributed/gloo in). */
#include "comd_host.h"
#include <string.h>
#include <time.h>
#include <stdlib.h>

static int myRank = 0;
static int nRanks = 1;
static CommTransport transport;
static int haveTransport = 0;
static int loopback = 0;        /* one rank that still talks through its transport (to itself): COMD_LOOPBACK_TRANSPORT=1 */

void initParallel(int rank, int n, const CommTransport* t)
{
   myRank = rank; nRanks = n;
   haveTransport = 0;
   if (t) { transport = *t; haveTransport = 1; }
   const char* lb = getenv("COMD_LOOPBACK_TRANSPORT");
   loopback = haveTransport && nRanks == 1 && lb && atoi(lb) != 0;
   if (nRanks > 1 && !haveTransport) {
      fprintf(stderr, "initParallel: %d ranks need a transport\n", nRanks);
      exit(-1);
   }
}

void destroyParallel(void) { haveTransport = 0; loopback = 0; myRank = 0; nRanks = 1; }

/* every message, reduction and broadcast goes through the transport even on one rank (exercises RCCL on a one-GPU box) */
int loopbackParallel(void) { return loopback; }
int getNRanks(void) { return nRanks; }
int getMyRank(void) { return myRank; }
int printRank(void) { return myRank == 0; }

void barrierParallel(void) { if (nRanks > 1 || loopback) transport.barrier(transport.ctx); }

void timestampBarrier(const char* msg)
{
   barrierParallel();
   if (!printRank()) return;
   time_t t = time(NULL);
   char* timeString = ctime(&t);
   timeString[24] = '\0';
   fprintf(screenOut, "%s: %s\n", timeString, msg);
   fflush(screenOut);
}

/* host buffers; single rank: copy to self, as the serial build of the reference does (parallel.c:112-117) */
int sendReceiveParallel(void* sendBuf, int sendLen, int dest, void* recvBuf, int recvLen, int source)
{
   if (nRanks == 1) { (void)dest; (void)source; if (sendLen > recvLen) sendLen = recvLen; memcpy(recvBuf, sendBuf, (size_t)sendLen); return sendLen; }
   return transport.sendrecv(transport.ctx, sendBuf, sendLen, dest, recvBuf, recvLen, source, 0, NULL);
}

/* device buffers, ordered on `stream` */
int sendReceiveDevice(void* sendBuf, int sendLen, int dest, void* recvBuf, int recvLen, int source, comdStream_t stream)
{
   if (nRanks == 1 && !loopback) { comdMemcpyDtoDAsync(recvBuf, sendBuf, sendLen, stream); return sendLen; }
   return transport.sendrecv(transport.ctx, sendBuf, sendLen, dest, recvBuf, recvLen, source, 1, stream);
}

/* both messages of one axis phase (device buffers).  Uses the transport's fused call when it has one. */
void sendReceiveDevice2(void* sendM, int nSendM, int dstM, void* recvP, void* sendP, int nSendP, int dstP, void* recvM,
                        int recvCap, comdStream_t stream, int nRecv[2])
{
   if (haveTransport && transport.sendrecv2 && (nRanks > 1 || loopback)) {
      transport.sendrecv2(transport.ctx, sendM, nSendM, dstM, recvP, sendP, nSendP, dstP, recvM, recvCap, 1, stream, nRecv);
      return;
   }
   nRecv[0] = sendReceiveDevice(sendM, nSendM, dstM, recvP, recvCap, dstP, stream);
   nRecv[1] = sendReceiveDevice(sendP, nSendP, dstP, recvM, recvCap, dstM, stream);
}

/* 1 when halo messages leave this rank through a transport that can move pre-agreed sizes without a handshake */
int sizedExchangeAvailable(void) { return haveTransport && transport.sendrecv2sized != NULL && (nRanks > 1 || loopback); }

void sendReceiveDevice2Sized(void* sendM, int nSendM, int dstM, void* recvP, int nRecvP, void* sendP, int nSendP, int dstP, void* recvM, int nRecvM,
                             comdStream_t stream)
{
   transport.sendrecv2sized(transport.ctx, sendM, nSendM, dstM, recvP, nRecvP, sendP, nSendP, dstP, recvM, nRecvM, 1, stream);
}

void addIntParallel(int* sendBuf, int* recvBuf, int count)
{
   memmove(recvBuf, sendBuf, (size_t)count * sizeof(int));
   if (nRanks > 1 || loopback) transport.allreduce(transport.ctx, recvBuf, count, 0);
}

void addRealParallel(real_t* sendBuf, real_t* recvBuf, int count)
{
   memmove(recvBuf, sendBuf, (size_t)count * sizeof(real_t));
   if (nRanks > 1 || loopback) transport.allreduce(transport.ctx, recvBuf, count, sizeof(real_t) == sizeof(double) ? 1 : 3);
}

void addDoubleParallel(double* sendBuf, double* recvBuf, int count)
{
   memmove(recvBuf, sendBuf, (size_t)count * sizeof(double));
   if (nRanks > 1 || loopback) transport.allreduce(transport.ctx, recvBuf, count, 1);
}

void maxIntParallel(int* sendBuf, int* recvBuf, int count)
{
   memmove(recvBuf, sendBuf, (size_t)count * sizeof(int));
   if (nRanks > 1 || loopback) transport.allreduce(transport.ctx, recvBuf, count, 2);
}

void bcastParallel(void* buf, int len, int root) { if (nRanks > 1 || loopback) transport.bcast(transport.ctx, buf, len, root); }
