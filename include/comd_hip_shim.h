/* comd_hip_shim.h -- reference-side adapter: lets the UNMODIFIED call expressions of the reference's host files
 * (src-mpi/ljForce.c:141, eam.c:203-259, timestep.c:137-160, :187, :224-275, haloExchange.c:1617-1633, :1686, :1872, :1885)
 * compile and link against libcomd_hip.so.
 *
 * A maintainer includes this header where the reference includes "gpu_kernels.h" / "gpu_utility.h" / <cuda_runtime.h>
 * (after the definition of SimFlat, which must embed `SimGpu gpu` from comd_hip.h -- CoMDTypes.h:75-135 does, through
 * gpu_types.h).  Everything below is preprocessor renaming plus a few static inline adapters for the entry points whose
 * reference signature carries host staging (the reference packs on the device, copies to a pinned host buffer and sends that
 * with MPI; libcomd_hip keeps messages on the device, so the adapters add the copy the call site expects).
 *
 * Compiled and linked by tests/test_boundary_shim.py (gcc, -Wall -Werror) against a mirror of the reference's SimFlat fields, and RUN on the GPU by the
 * same file (tests/shim/shim_driver.c drives whole time steps through these adapters); nothing in the product includes this file.
 *
 * Function-like macros with the name of the function they wrap are deliberate: inside its own expansion a macro name is not
 * expanded again, and `(name)(...)` always reaches the real function.
 */
#ifndef COMD_HIP_SHIM_H
#define COMD_HIP_SHIM_H

#include <limits.h>
#include "comd_hip.h"

/* ---- CUDA runtime names used by the hot-path host files (ljForce.c, eam.c, timestep.c, haloExchange.c + the headers they include) ------------------ */
#include <stddef.h>
typedef comdStream_t cudaStream_t;
typedef void* cudaEvent_t;                              /* haloExchange.h:132 (a field of the libmp exchange state; never used on the plain path) */
typedef int cudaError_t;                                /* libcomd_hip reports HIP errors itself (message + exit(-1), as CUDA_CHECK does): calls here always succeed */
enum { cudaSuccess = 0 };
enum { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3 };
enum { cudaHostRegisterDefault = 0, cudaHostRegisterPortable = 1, cudaHostRegisterMapped = 2 };
static inline const char* cudaGetErrorString(cudaError_t e) { (void)e; return "no error (libcomd_hip exits with the HIP error string on its own)"; }
static inline cudaError_t comdShimOk(void) { return cudaSuccess; }
#define cudaStreamSynchronize(s)              (comdStreamSynchronize(s), comdShimOk())
#define cudaDeviceSynchronize()               (comdDeviceSynchronize(), comdShimOk())
#define cudaMemset(p, v, n)                   (comdDeviceMemset((p), (v), (long)(n)), comdShimOk())
#define cudaMemcpyAsync(dst, src, n, kind, s) (comdMemcpyAsync((dst), (src), (long)(n), (kind), (s)), comdShimOk())
#define cudaMemcpy(dst, src, n, kind)         (comdMemcpyAsync((dst), (src), (long)(n), (kind), (comdStream_t)0), comdStreamSynchronize((comdStream_t)0), comdShimOk())
/* haloExchange.c:229-266, 295-308, 384-467 (buffers and cell lists of the exchange), :1698-1701, :1908-1914 */
static inline cudaError_t comdShimMalloc(void** p, size_t n)     { *p = comdDeviceMalloc((long)n); return cudaSuccess; }
static inline cudaError_t comdShimMallocHost(void** p, size_t n) { *p = comdHostMallocPinned((long)n); return cudaSuccess; }
#define cudaMalloc(p, n)            comdShimMalloc((void**)(p), (size_t)(n))
#define cudaMallocHost(p, n)        comdShimMallocHost((void**)(p), (size_t)(n))
#define cudaFree(p)                 (comdDeviceFree((void*)(p)), comdShimOk())
#define cudaFreeHost(p)             (comdHostFreePinned((void*)(p)), comdShimOk())
#define cudaHostRegister(p, n, f)   comdShimOk()       /* haloExchange.c:215-218: the host buffers stay pageable; comdMemcpyAsync copies them either way */

/* nvToolsExt.h as haloExchange.c:60-91 uses it (PUSH_RANGE / POP_RANGE around the exchange phases): profiler range markers, no effect on the computation.
 * rocprofv3 sees the kernels by name; the markers compile to nothing. */
typedef struct { int version, size, colorType; unsigned color; int messageType; struct { const char* ascii; } message; } nvtxEventAttributes_t;
enum { NVTX_VERSION = 0, NVTX_EVENT_ATTRIB_STRUCT_SIZE = 0, NVTX_COLOR_ARGB = 0, NVTX_MESSAGE_TYPE_ASCII = 0 };
static inline int nvtxRangePushEx(const nvtxEventAttributes_t* a) { (void)a; return 0; }
static inline int nvtxRangePop(void) { return 0; }

/* MAXATOMS is a -D macro in the reference (Makefile:16); here it is a run-time value.  Usable inside functions that have the
 * simulation in scope under the reference's usual names (`sim` or `s`): define COMD_SHIM_SIM to that name before including. */
#ifdef COMD_SHIM_SIM
#define MAXATOMS ((COMD_SHIM_SIM)->gpu.maxAtoms)
#endif

/* mytype.h:30 */
typedef real_t real3_old[3];
/* gpu_types.h:115-121: a_list / i_list / b_list have no counterpart (kernels address cell slots); the arguments that name them
 * are dropped by the macros below without being evaluated */

/* ---- force: gpu_kernels.h:13-24 -------------------------------------------------------------------------------------- */
/* ljForceGpu(&(sim->gpu), interpolation, nLocalBoxes, NULL, plcutoff, method): identical signature (ljForce.c:141) */
#define eamForce1Gpu(g, method, spline) (eamForce1Gpu)(&(g), (method), (spline))                    /* eam.c:232 */
#define eamForce2Gpu(g, method, spline) (eamForce2Gpu)(&(g), (method), (spline))                    /* eam.c:235 */
#define eamForce3Gpu(g, method, spline) (eamForce3Gpu)(&(g), (method), (spline))                    /* eam.c:259 */
#define eamForce1GpuAsync(g, list, n, cells, method, stream, spline) (eamForce1GpuAsync)(&(g), (n), (cells), (method), (stream), (spline))   /* eam.c:207, timestep.c:260 */
#define eamForce2GpuAsync(g, list, n, cells, method, stream, spline) (eamForce2GpuAsync)(&(g), (n), (cells), (method), (stream), (spline))   /* eam.c:208, timestep.c:261 */
#define eamForce3GpuAsync(g, list, n, cells, method, stream, spline) (eamForce3GpuAsync)(&(g), (n), (cells), (method), (stream), (spline))   /* eam.c:215, 255 */
#define updateNeighborsGpu(g, temp) (updateNeighborsGpu)(&(g), (temp))                              /* eam.c:220 */
#define updateNeighborsGpuAsync(g, temp, n, cells, stream) (updateNeighborsGpuAsync)(&(g), (temp), (n), (cells), (stream))   /* eam.c:204, timestep.c:257 */

/* ---- integrator + energy: gpu_kernels.h:79-83 ------------------------------------------------------------------------ */
#define advanceVelocityGpu(g, dt) (advanceVelocityGpu)(&(g), (dt))                                  /* timestep.c:138 */
/* advancePositionGpu(&(s->gpu), dt): identical signature (timestep.c:158) */
#define computeEnergy(sim, eLocal) (computeEnergy)(&(sim)->gpu, (eLocal))                           /* timestep.c:188 */

/* ---- redistribute: gpu_kernels.h:84-86.  The reference keeps its streams in SimFlat (CoMDTypes.h:116-117) -------------- */
/* the reference's updateLinkCellsGpu returns after a blocking device read (gpu_kernels.cu:497), and its callers rely on that: timestep.c:257-265 starts
 * the interior cells' force work on ANOTHER stream right after it.  The library's call only enqueues, so the adapter adds the wait. */
#define updateLinkCellsGpu(sim)       ((updateLinkCellsGpu)(&(sim)->gpu, (sim)->boundary_stream), comdStreamSynchronize((sim)->boundary_stream))     /* timestep.c:234, 246 */
#define buildAtomListGpu(sim, stream) (buildAtomListGpu)(&(sim)->gpu, (stream))                     /* timestep.c:244, 271 */
#define emptyHaloCellsGpu(sim)        (emptyHaloCellsGpu)(&(sim)->gpu, (sim)->boundary_stream)       /* timestep.c:282, 340 */
#define sortAtomsGpu(sim, stream)     (sortAtomsGpu)(&(sim)->gpu, (stream))                         /* timestep.c:248, 274 */

/* ---- halo pack / unpack: gpu_kernels.h:28, 70-72 ----------------------------------------------------------------------
 * The reference's device staging buffers (SimFlat.gpu_atoms_buf / gpu_force_buf, CoMDTypes.h:125-126) hold the bare SoA payload.
 * libcomd_hip puts a 16-byte count header in front of an atom message: allocate gpu_atoms_buf with
 * comdShimAtomsBufAlloc(bytes), which returns a pointer 16 bytes into a larger allocation, so that the header lands in front of
 * the address the reference code knows and its cudaMemcpyAsync of nAtoms * sizeof(AtomMsg) bytes moves exactly the payload. */
static inline char* comdShimAtomsBufAlloc(long payloadBytes) { return (char*)comdDeviceMalloc(payloadBytes + COMD_ATOM_MSG_HEADER) + COMD_ATOM_MSG_HEADER; }
static inline void  comdShimAtomsBufFree(char* p) { if (p) comdDeviceFree(p - COMD_ATOM_MSG_HEADER); }

/* scan scratch of nCells + 1 ints (the reference's partial_sums arrays hold nCells): grown on demand, kept in the SimGpu it serves
 * (SimGpu.adapterScan; two simulations on two devices never share it), freed by DestroyGpu */
static inline int* comdShimOffsets(SimGpu* gpu, int nCells)
{
   if (nCells + 1 > gpu->adapterScanCap) {
      if (gpu->adapterScan) comdDeviceFree(gpu->adapterScan);
      gpu->adapterScanCap = 2 * (nCells + 1);
      gpu->adapterScan = (int*)comdDeviceMalloc((long)gpu->adapterScanCap * (long)sizeof(int));
   }
   return gpu->adapterScan;
}

/* int compactCellsGpu(work_d, nCells, d_cellList, SimGpu, d_cellOffsets, d_workScan, shift, stream), gpu_kernels.cu:519-551:
 * returns the number of atoms packed (a blocking read, as in the reference :534-535) */
static inline int comdShimCompactCells(char* work_d, int nCells, int* d_cellList, SimGpu* gpu, int* d_cellOffsets, const real_t* shift, comdStream_t stream)
{
   (void)d_cellOffsets;                                  /* nCells ints in the reference; the scan here needs nCells + 1 */
   char* msg = work_d - COMD_ATOM_MSG_HEADER;
   compactCellsGpu(msg, nCells, d_cellList, gpu, comdShimOffsets(gpu, nCells), shift, INT_MAX, stream);
   return atomMsgCountGpu(gpu, msg, stream);
}
#define compactCellsGpu(work_d, nCells, d_cellList, g, d_cellOffsets, d_workScan, shift, stream) \
   comdShimCompactCells((work_d), (nCells), (d_cellList), &(g), (d_cellOffsets), (shift), (stream))     /* haloExchange.c:1617 */

/* void unloadAtomsBufferToGpu(buf (host), nBuf, SimFlat*, gpu_buf, stream), gpu_kernels.cu:572-617: copy the received payload to
 * the device staging buffer, then bin the atoms by coordinate */
static inline void comdShimUnloadAtoms(const char* hostBuf, int nBuf, SimGpu* gpu, char* gpu_buf, comdStream_t stream)
{
   comdMemcpyAsync(gpu_buf, hostBuf, (long)nBuf * COMD_ATOM_MSG_BYTES_PER_ATOM, cudaMemcpyHostToDevice, stream);
   unloadAtomsBufferToGpu(gpu_buf - COMD_ATOM_MSG_HEADER, nBuf, nBuf, gpu, stream);
}
#define unloadAtomsBufferToGpu(buf, nBuf, sim, gpu_buf, stream) comdShimUnloadAtoms((buf), (nBuf), &(sim)->gpu, (gpu_buf), (stream))   /* haloExchange.c:1686 */

/* void loadForceBufferFromGpu(buf (host), &nBuf, nCells, cellList, natoms_buf, partial_sums, SimFlat*, gpu_buf, stream), :619-640 */
static inline void comdShimLoadForce(char* hostBuf, int* nBuf, int nCells, int* d_cellList, SimGpu* gpu, char* gpu_buf, comdStream_t stream)
{
   int* off = comdShimOffsets(gpu, nCells);
   loadForceBufferFromGpu((real_t*)gpu_buf, nCells, d_cellList, off, gpu, stream);
   *nBuf = comdReadDeviceInt(off + nCells, stream);
   comdMemcpyAsync(hostBuf, gpu_buf, (long)*nBuf * (long)sizeof(real_t), cudaMemcpyDeviceToHost, stream);
   comdStreamSynchronize(stream);
}
#define loadForceBufferFromGpu(buf, nbuf, nCells, cellList, natoms_buf, partial_sums, s, gpu_buf, stream) \
   comdShimLoadForce((buf), (nbuf), (nCells), (cellList), &(s)->gpu, (gpu_buf), (stream))               /* haloExchange.c:1872 */

/* void unloadForceBufferToGpu(buf (host), nBuf, nCells, cellList, natoms_buf, partial_sums, SimFlat*, gpu_buf, stream), :642-660 */
static inline void comdShimUnloadForce(const char* hostBuf, int nBuf, int nCells, int* d_cellList, SimGpu* gpu, char* gpu_buf, comdStream_t stream)
{
   comdMemcpyAsync(gpu_buf, hostBuf, (long)nBuf * (long)sizeof(real_t), cudaMemcpyHostToDevice, stream);
   unloadForceBufferToGpu((const real_t*)gpu_buf, nCells, d_cellList, comdShimOffsets(gpu, nCells), gpu, stream);
}
#define unloadForceBufferToGpu(buf, nBuf, nCells, cellList, natoms_buf, partial_sums, s, gpu_buf, stream) \
   comdShimUnloadForce((buf), (nBuf), (nCells), (cellList), &(s)->gpu, (gpu_buf), (stream))             /* haloExchange.c:1885 */

/* ---- device management: gpu_utility.h:55-69.  AllocateGpu / SetBoundaryCells / CopyDataToGpu / GetDataFromGpu take SimFlat* in the
 * reference and read its link cells, potential and halo lists; here they take a GpuConfig / explicit lists / HostAtoms
 * (INTEGRATION.md section 2 shows the three short call-site rewrites) -- no macro can invent those arguments. --------------------- */

#endif
