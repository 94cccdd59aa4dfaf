/* comd_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp64) of the CoMD hot path of e-ago/CoMD-CUDA-Async:
 * FCC/temperature/displacement initialisation, link-cell redistribution,
 * 6-face halo exchange, LJ and EAM forces, velocity-Verlet time step.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link,
 * load or call this library.  The product path (comd-cuda-async_amd/) never does.
 *
 * Parity pins (see DESIGN.md "Oracle"): CoMD.c:896-900 (EAM cohesive energy),
 * the step-0 row of errors_sync_version/SyncVersion_error16nodes/out16_80_3.txt, and
 * the values SURVEY.md section 8c recorded from the unmodified reference.
 */
#ifndef COMD_ORACLE_H
#define COMD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleSim OracleSim;

/* which-array selectors for oracle_gather / oracle_scatter / oracle_rank_array */
enum { ORACLE_R = 0, ORACLE_P = 1, ORACLE_F = 2, ORACLE_U = 3, ORACLE_RHOBAR = 4, ORACLE_DFEMBED = 5 };

/* Build a simulation exactly as the reference's initSimulation does (CoMD.c:200-327):
 * nx,ny,nz unit cells; px,py,pz virtual ranks (all simulated inside this process);
 * lat<0 -> the potential's lattice constant; doeam 0 = LJ, 1 = EAM funcfl file potDir/potName;
 * cellCap = link-cell capacity (MAXATOMS of the reference Makefile; 0 -> 64 for EAM, 512 for LJ).
 * Performs the initial redistribute + force + kinetic energy.  Returns NULL on error. */
OracleSim* oracle_create(int nx, int ny, int nz, int px, int py, int pz,
                         double lat, int doeam, const char* potDir, const char* potName,
                         double temperature, double initialDelta, double dt, int cellCap);
/* test-only: LJ cutoff in sigmas for the simulations created afterwards (default 5 = ljForce.c:114; 2.5 meets CoMD.c:897) */
void oracle_set_lj_cutoff_sigmas(double f);
void oracle_destroy(OracleSim* s);

/* timestep() of timestep.c:48-100: nSteps velocity-Verlet steps, then kinetic energy. */
void oracle_step(OracleSim* s, int nSteps);
/* individual phases, for kernel-level parity */
void oracle_advance_velocity(OracleSim* s, double dt);
void oracle_advance_position(OracleSim* s, double dt);
void oracle_redistribute(OracleSim* s);
void oracle_compute_force(OracleSim* s);
/* EAM only: switch phi and rho to the reference's GPU-only cubic splines in r^2 (-P) and recompute the forces.  PARITY-UNPINNED:
 * the reference holds no output of this mode and its CPU path does not implement it. */
void oracle_use_splines(OracleSim* s);
void oracle_kinetic_energy(OracleSim* s);

int    oracle_n_global(const OracleSim* s);
int    oracle_n_ranks(const OracleSim* s);
double oracle_e_potential(const OracleSim* s);   /* total, eV */
double oracle_e_kinetic(const OracleSim* s);     /* total, eV */
double oracle_cutoff(const OracleSim* s);
double oracle_mass(const OracleSim* s);
double oracle_lattice(const OracleSim* s);

/* per-atom arrays of LOCAL atoms of all ranks, indexed by gid:
 * R,P,F -> out[3*gid+{0,1,2}]; U, RHOBAR, DFEMBED -> out[gid].  Positions are the
 * owning rank's coordinates (inside the global box). */
void oracle_gather(const OracleSim* s, int which, double* out);
/* overwrite R or P of every local atom from a by-gid array (then call
 * oracle_redistribute + oracle_compute_force). */
void oracle_scatter(OracleSim* s, int which, const double* in);

/* link-cell view of one virtual rank (bit-exact checks of index work) */
int  oracle_rank_cell_cap(const OracleSim* s);
void oracle_rank_grid(const OracleSim* s, int rank, int gridSize[3], int* nLocalBoxes, int* nTotalBoxes);
void oracle_rank_natoms(const OracleSim* s, int rank, int* nAtoms /* [nTotalBoxes] */);
void oracle_rank_gid(const OracleSim* s, int rank, int* gid /* [nTotalBoxes*cap] */);
void oracle_rank_array(const OracleSim* s, int rank, int which, int comp, double* out /* [nTotalBoxes*cap] */);
/* geometry helpers restating linkCells.c */
int  oracle_box_from_tuple(const OracleSim* s, int rank, int ix, int iy, int iz);
int  oracle_box_from_coord(const OracleSim* s, int rank, const double r[3]);
/* halo cell lists of haloExchange.c:1543-1567 / 1712-1801; kind 0 = atom list, 1 = force send, 2 = force recv.
 * Returns the number of cells; list may be NULL to query the size. */
int  oracle_face_cells(const OracleSim* s, int rank, int kind, int face, int* list);

/* building blocks exposed for pinning against oracle/_ref (reference random.c) and the tables */
double   oracle_lcg61(uint64_t* seed);
uint64_t oracle_mkSeed(uint32_t id, uint32_t callSite);
double   oracle_gasdev(uint64_t* seed);
/* table: 0 = phi, 1 = rho, 2 = F */
int      oracle_eam_interpolate(const OracleSim* s, int table, double x, double* f, double* df);
int      oracle_eam_table(const OracleSim* s, int table, int* n, double* x0, double* invDx, double* values /* n+3, may be NULL */);

/* wall-clock seconds spent inside oracle_step's loop since creation (cpu_baseline) */
double oracle_loop_seconds(const OracleSim* s);
/* number of OpenMP threads the force loops use (1 when built without -fopenmp) */
int    oracle_threads(void);
void   oracle_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
