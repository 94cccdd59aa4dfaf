/* eam_potential.c -- EAM potential plugin: funcfl table reader (eam.c:802-872), interpolation tables with the
 * reference's padding rule (:496-519) and host evaluation (:557-579), report (:421-431), and the three-pass force
 * choreography with the dfEmbed halo exchange in the middle (eamForceGpu, eam.c:196-264). */
#include "comd_host.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int eamForce(SimFlat* s);
static void eamPrint(FILE* file, BasePotential* pot);
static void eamDestroy(BasePotential** pot);

static InterpolationObject* initInterpolationObject(int n, real_t x0, real_t dx, const real_t* data)
{
   InterpolationObject* t = (InterpolationObject*)calloc(1, sizeof(InterpolationObject));
   real_t* raw = (real_t*)calloc((size_t)n + 3, sizeof(real_t));
   t->values = raw + 1;
   t->n = n; t->invDx = 1.0 / dx; t->x0 = x0; t->invDxXx0 = x0 * t->invDx;
   for (int i = 0; i < n; ++i) t->values[i] = data[i];
   t->values[-1] = t->values[0];
   t->values[n + 1] = t->values[n] = t->values[n - 1];
   return t;
}

static void destroyInterpolationObject(InterpolationObject** a)
{
   if (!a || !*a) return;
   if ((*a)->values) free((*a)->values - 1);
   free(*a);
   *a = NULL;
}

void interpolate(InterpolationObject* table, real_t r, real_t* f, real_t* df)
{
   const real_t* tt = table->values;
   if (r < table->x0) r = table->x0;
   r = (r - table->x0) * table->invDx;
   int ii = (int)floor(r);
   if (ii > table->n) { ii = table->n; r = table->n / table->invDx; }
   r = r - floor(r);
   real_t g1 = tt[ii + 1] - tt[ii - 1];
   real_t g2 = tt[ii + 2] - tt[ii];
   *f = tt[ii] + 0.5 * r * (g1 + r * (tt[ii + 1] + tt[ii - 1] - 2.0 * tt[ii]));
   *df = 0.5 * (g1 + r * (g2 - g1)) * table->invDx;
}

static void fileNotFound(const char* callSite, const char* filename)
{
   fprintf(screenOut, "%s: Can't open file %s.  Fatal Error.\n", callSite, filename);
   exit(-1);
}

/* funcfl: line 1 comment (element first), line 2 "Z mass lat lattice", line 3 "nRho dRho nR dR cutoff", then F(rho),
 * Z(r) (converted to phi = Z^2/r in eV, phi[0] extrapolated) and rho(r). */
static void eamReadFuncfl(EamPotential* pot, const char* dir, const char* potName)
{
   char tmp[4096];
   snprintf(tmp, sizeof tmp, "%s/%s", dir, potName);
   FILE* fp = fopen(tmp, "r");
   if (!fp) fileNotFound("eamReadFuncfl", tmp);
   char name[16] = "";
   if (!fgets(tmp, sizeof tmp, fp)) fileNotFound("eamReadFuncfl", potName);
   sscanf(tmp, "%15s", name);
   strncpy(pot->name, name, 2); pot->name[2] = '\0';
   int nAtomic; double mass, lat; char latticeType[8];
   if (!fgets(tmp, sizeof tmp, fp)) fileNotFound("eamReadFuncfl", potName);
   sscanf(tmp, "%d %le %le %7s", &nAtomic, &mass, &lat, latticeType);
   pot->atomicNo = nAtomic; pot->lat = lat; pot->mass = mass * amuToInternalMass;
   strcpy(pot->latticeType, latticeType);
   int nRho, nR; double dRho, dR, cutoff;
   if (!fgets(tmp, sizeof tmp, fp)) fileNotFound("eamReadFuncfl", potName);
   sscanf(tmp, "%d %le %d %le %le", &nRho, &dRho, &nR, &dR, &cutoff);
   pot->cutoff = cutoff;
   const real_t x0 = 0.0;
   int bufSize = nRho > nR ? nRho : nR;
   real_t* buf = (real_t*)malloc((size_t)bufSize * sizeof(real_t));
   for (int i = 0; i < nRho; ++i) if (fscanf(fp, FMT1, buf + i) != 1) fileNotFound("eamReadFuncfl(F)", potName);
   pot->f = initInterpolationObject(nRho, x0, dRho, buf);
   for (int i = 0; i < nR; ++i) if (fscanf(fp, FMT1, buf + i) != 1) fileNotFound("eamReadFuncfl(Z)", potName);
   for (int i = 1; i < nR; ++i) {
      real_t r = x0 + i * dR;
      buf[i] *= buf[i] / r;
      buf[i] *= hartreeToEv * bohrToAngs;
   }
   buf[0] = buf[1] + (buf[1] - buf[2]);
   pot->phi = initInterpolationObject(nR, x0, dR, buf);
   for (int i = 0; i < nR; ++i) if (fscanf(fp, FMT1, buf + i) != 1) fileNotFound("eamReadFuncfl(rho)", potName);
   pot->rho = initInterpolationObject(nR, x0, dR, buf);
   free(buf);
   fclose(fp);
}

/* setfl (eam.c:680-757), one element: 3 comment lines, "ntypes ...", "nRho dRho nR dR cutoff", "Z mass lat lattice", then F(rho),
 * rho(r) and r*phi(r) (converted to phi, phi[0] extrapolated). */
static void eamReadSetfl(EamPotential* pot, const char* dir, const char* potName)
{
   char tmp[4096];
   snprintf(tmp, sizeof tmp, "%s/%s", dir, potName);
   FILE* fp = fopen(tmp, "r");
   if (!fp) fileNotFound("eamReadSetfl", tmp);
   for (int i = 0; i < 3; ++i) if (!fgets(tmp, sizeof tmp, fp)) fileNotFound("eamReadSetfl", potName);
   int nElems = 0;
   if (!fgets(tmp, sizeof tmp, fp)) fileNotFound("eamReadSetfl", potName);
   sscanf(tmp, "%d", &nElems);
   if (nElems != 1) {
      fprintf(screenOut, "eamReadSetfl: CoMD 1.1 does not support alloys and cannot\n   read setfl files with multiple species.  Fatal Error.\n");
      exit(-1);
   }
   int nRho, nR; double dRho, dR, cutoff;
   if (!fgets(tmp, sizeof tmp, fp)) fileNotFound("eamReadSetfl", potName);
   sscanf(tmp, "%d %le %d %le %le", &nRho, &dRho, &nR, &dR, &cutoff);
   pot->cutoff = cutoff;
   int nAtomic; double mass, lat; char latticeType[8];
   if (!fgets(tmp, sizeof tmp, fp)) fileNotFound("eamReadSetfl", potName);
   sscanf(tmp, "%d %le %le %7s", &nAtomic, &mass, &lat, latticeType);
   pot->atomicNo = nAtomic; pot->lat = lat; pot->mass = mass * amuToInternalMass;
   strcpy(pot->latticeType, latticeType);
   strcpy(pot->name, "Cu");                     /* the reference leaves name unset for setfl; the only shipped file is Cu */
   const real_t x0 = 0.0;
   int bufSize = nRho > nR ? nRho : nR;
   real_t* buf = (real_t*)malloc((size_t)bufSize * sizeof(real_t));
   for (int i = 0; i < nRho; ++i) if (fscanf(fp, FMT1, buf + i) != 1) fileNotFound("eamReadSetfl(F)", potName);
   pot->f = initInterpolationObject(nRho, x0, dRho, buf);
   for (int i = 0; i < nR; ++i) if (fscanf(fp, FMT1, buf + i) != 1) fileNotFound("eamReadSetfl(rho)", potName);
   pot->rho = initInterpolationObject(nR, x0, dR, buf);
   for (int i = 0; i < nR; ++i) if (fscanf(fp, FMT1, buf + i) != 1) fileNotFound("eamReadSetfl(phi)", potName);
   for (int i = 1; i < nR; ++i) buf[i] /= (x0 + i * dR);
   buf[0] = buf[1] + (buf[1] - buf[2]);
   pot->phi = initInterpolationObject(nR, x0, dR, buf);
   free(buf);
   fclose(fp);
}

BasePotential* initEamPot(const char* dir, const char* file, const char* type)
{
   EamPotential* pot = (EamPotential*)calloc(1, sizeof(EamPotential));
   pot->force = eamForce;
   pot->print = eamPrint;
   pot->destroy = eamDestroy;
   /* every rank reads the (36 kB) file itself; the reference reads on rank 0 and broadcasts (eam.c:160-171) */
   if (strcmp(type, "funcfl") == 0) eamReadFuncfl(pot, dir, file);
   else if (strcmp(type, "setfl") == 0) eamReadSetfl(pot, dir, file);
   else {
      fprintf(screenOut, "initEamPot: Potential type %s not supported. Fatal Error.\n", type);
      exit(-1);
   }
   return (BasePotential*)pot;
}

static void eamPrint(FILE* file, BasePotential* pot)
{
   EamPotential* e = (EamPotential*)pot;
   fprintf(file, "  Potential type  : EAM\n");
   fprintf(file, "  Species name    : %s\n", e->name);
   fprintf(file, "  Atomic number   : %d\n", e->atomicNo);
   fprintf(file, "  Mass            : %lg amu\n", e->mass / amuToInternalMass);
   fprintf(file, "  Lattice type    : %s\n", e->latticeType);
   fprintf(file, "  Lattice spacing : %lg Angstroms\n", e->lat);
   fprintf(file, "  Cutoff          : %lg Angstroms\n", e->cutoff);
}

/* gpu_utility.c:377-430 initSplineCoefficients: natural cubic spline (Numerical Recipes 3.3) through the table samples as a function
 * of x = r^2 -- second derivative 0 at the first knot, first derivative 0 at the last -- returned as {a,b,c,d} per interval so that
 * f = ((a x + b) x + c) x + d.  values[0..n] are read (values[n] is the table's trailing pad). */
real_t* initSplineCoefficients(int n, const real_t* values, real_t x0, real_t invDx)
{
   real_t* u = (real_t*)malloc((size_t)n * sizeof(real_t));
   real_t* d2 = (real_t*)malloc((size_t)(n + 1) * sizeof(real_t));
   d2[0] = 0.0; u[0] = 0.0;
   for (int i = 1; i < n; ++i) {
      const real_t xi = (x0 + i / invDx) * (x0 + i / invDx);
      const real_t xp = (x0 + (i - 1) / invDx) * (x0 + (i - 1) / invDx);
      const real_t xn = (x0 + (i + 1) / invDx) * (x0 + (i + 1) / invDx);
      const real_t sig = (xi - xp) / (xn - xp);
      const real_t p = sig * d2[i - 1] + 2.0;
      d2[i] = (sig - 1.0) / p;
      u[i] = (values[i + 1] - values[i]) / (xn - xi) - (values[i] - values[i - 1]) / (xi - xp);
      u[i] = (6.0 * u[i] / (xn - xp) - sig * u[i - 1]) / p;
   }
   {
      const real_t xn = (x0 + n / invDx) * (x0 + n / invDx);
      const real_t xnp = (x0 + (n - 1) / invDx) * (x0 + (n - 1) / invDx);
      const real_t qn = 0.5;
      const real_t un = (-3.0 / (xn - xnp)) * (values[n] - values[n - 1]) / (xn - xnp);
      d2[n] = (un - qn * u[n - 1]) / (qn * d2[n - 1] + 1.0);
   }
   for (int i = n - 1; i >= 0; --i) d2[i] = d2[i] * d2[i + 1] + u[i];
   real_t* c = (real_t*)malloc((size_t)4 * n * sizeof(real_t));
   for (int i = 0; i < n; ++i) {
      const real_t x1 = (x0 + i / invDx) * (x0 + i / invDx);
      const real_t x2 = (x0 + (i + 1) / invDx) * (x0 + (i + 1) / invDx);
      const real_t h = x2 - x1, s1 = d2[i], s2 = d2[i + 1], y1 = values[i], y2 = values[i + 1];
      c[4 * i]     = 1.0 / (6.0 * h) * (s2 - s1);
      c[4 * i + 1] = 1.0 / (2.0 * h) * (x2 * s1 - x1 * s2);
      c[4 * i + 2] = 1.0 / h * (1.0 / 6.0 * (-3 * x2 * x2 + h * h) * s1 + 1.0 / 6.0 * (3 * x1 * x1 - h * h) * s2 - y1 + y2);
      c[4 * i + 3] = 1 / h * (x2 * y1 - x1 * y2 + 1.0 / 6.0 * s1 * (x2 * x2 * x2 - x2 * h * h) + 1.0 / 6.0 * s2 * (-x1 * x1 * x1 + x1 * h * h));
   }
   free(u); free(d2);
   return c;
}

void eamUseSplines(BasePotential* bp)
{
   EamPotential* pot = (EamPotential*)bp;
   pot->phiSpline = initSplineCoefficients(pot->phi->n, pot->phi->values, pot->phi->x0, pot->phi->invDx);
   pot->rhoSpline = initSplineCoefficients(pot->rho->n, pot->rho->values, pot->rho->x0, pot->rho->invDx);
}

static void eamDestroy(BasePotential** pPot)
{
   if (!pPot || !*pPot) return;
   EamPotential* pot = (EamPotential*)*pPot;
   destroyInterpolationObject(&pot->phi);
   destroyInterpolationObject(&pot->rho);
   destroyInterpolationObject(&pot->f);
   free(pot->phiSpline); free(pot->rhoSpline);
   if (pot->forceExchange) destroyHaloExchange(&pot->forceExchange);
   free(pot);
   *pPot = NULL;
}

static int eamForce(SimFlat* s)
{
   EamPotential* pot = (EamPotential*)s->pot;
   SimGpu* g = &s->gpu;
   if (s->gpuAsync) {
      /* passes 1-2 of the interior cells were launched on interior_stream by redistributeAtoms (timestep.c:257-265) */
      ensureInteriorForceLaunched(s);
      eamForce1GpuAsync(g, s->n_boundary_cells, g->boundary_cells, s->method, g->boundary_stream, s->spline);
      eamForce2GpuAsync(g, s->n_boundary_cells, g->boundary_cells, s->method, g->boundary_stream, s->spline);
      comdStreamSynchronize(g->boundary_stream);       /* boundary dfEmbed must exist before it is packed */
      comdStreamSynchronize(g->interior_stream);       /* pass 3 of interior cells reads dfEmbed of ring-2 boundary cells */
      eamForce3GpuAsync(g, g->n_interior_cells, g->interior_cells, s->method, g->interior_stream, s->spline);
   } else {
      eamForce1Gpu(g, s->method, s->spline);
      if (!s->gpuProfile) eamForce2Gpu(g, s->method, s->spline);
   }
   if (!s->gpuProfile) {
      startTimer(eamHaloTimer);
      prepareForceExchange(pot->forceExchange, s);
      comdForceScansReady(g, 1);
      haloExchange(pot->forceExchange, s);
      comdForceScansReady(g, 0);
      stopTimer(eamHaloTimer);
      if (s->gpuAsync) {
         eamForce3GpuAsync(g, s->n_boundary_cells, g->boundary_cells, s->method, g->boundary_stream, s->spline);
         comdDeviceSynchronize();
      } else {
         eamForce3Gpu(g, s->method, s->spline);
      }
   }
   return 0;
}
