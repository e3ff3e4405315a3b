/* ref_metis_main.c -- tiny driver around the reference's vendored METIS 5.1.0
 * (/root/reference/SuiteSparse/metis-5.1.0, compiled unmodified by oracle/Makefile).
 * TEST INFRASTRUCTURE ONLY.  Produces the element partition epart[] exactly as the reference's
 * wrapper does (src/Utils/METIS.hpp:83-140 builds eptr/eind, :265-300 sets the options:
 * k-way, cut objective, SHEM coarsening, METIS-RB initial partition, greedy refinement,
 * minconn=1, contig=1, ncuts=3, nseps=3, niter=10, ncommon=3, seed=-1, ufactor=30, dbglvl 511
 * -- dbglvl only prints, we pass 0).
 *
 * usage: metis_part <tets.i32 (nT*4 raw int32)> <nV> <nParts> <out epart.i32> [nodal]
 * nodal: METIS::partMesh_nodes (src/Utils/METIS.hpp:161-193, METIS_PartMeshNodal, unit vertex weights) -- the VERTEX
 *        partition LBFGS-JH works on (LBFGSTimeStepper.cpp:70-90); the output is then npart (nV raw int32)
 */
#include <metis.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s tets.i32 nV nParts out.i32\n", argv[0]);
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    idx_t ne = bytes / 16, nn = atol(argv[2]), nparts = atol(argv[3]);
    int32_t *T = malloc(bytes);
    if (fread(T, 1, bytes, f) != (size_t)bytes) return 3;
    fclose(f);

    idx_t *eptr = malloc(sizeof(idx_t) * (ne + 1)), *eind = malloc(sizeof(idx_t) * ne * 4);
    for (idx_t e = 0; e < ne; ++e) {
        eptr[e] = e * 4;
        for (int k = 0; k < 4; ++k) eind[4 * e + k] = T[4 * e + k];
    }
    eptr[ne] = ne * 4;

    idx_t options[METIS_NOPTIONS];
    METIS_SetDefaultOptions(options);
    options[METIS_OPTION_PTYPE] = METIS_PTYPE_KWAY;
    options[METIS_OPTION_OBJTYPE] = METIS_OBJTYPE_CUT;
    options[METIS_OPTION_CTYPE] = METIS_CTYPE_SHEM;
    options[METIS_OPTION_IPTYPE] = METIS_IPTYPE_METISRB;
    options[METIS_OPTION_RTYPE] = METIS_RTYPE_GREEDY;
    options[METIS_OPTION_MINCONN] = 1;
    options[METIS_OPTION_CONTIG] = 1;
    options[METIS_OPTION_NCUTS] = 3;
    options[METIS_OPTION_NSEPS] = 3;
    options[METIS_OPTION_NITER] = 10;
    options[METIS_OPTION_DBGLVL] = 0;
    options[METIS_OPTION_SEED] = -1;
    options[METIS_OPTION_UFACTOR] = 30;

    idx_t *epart = malloc(sizeof(idx_t) * ne), *npart = malloc(sizeof(idx_t) * nn);
    idx_t *ewgt = malloc(sizeof(idx_t) * ne);
    real_t *tpwgts = malloc(sizeof(real_t) * nparts);
    for (idx_t e = 0; e < ne; ++e) ewgt[e] = 1;
    for (idx_t p = 0; p < nparts; ++p) tpwgts[p] = 1.0 / nparts;
    idx_t ncommon = 3, objval = 0;
    const int nodal = argc > 5 && argv[5][0] == 'n';
    int status;
    if (nodal) {
        idx_t *vwgt = malloc(sizeof(idx_t) * nn);
        for (idx_t v = 0; v < nn; ++v) vwgt[v] = 1;
        status = METIS_PartMeshNodal(&ne, &nn, eptr, eind, vwgt, NULL, &nparts, tpwgts, options, &objval, epart, npart);
    } else {
        status = METIS_PartMeshDual(&ne, &nn, eptr, eind, ewgt, NULL, &ncommon, &nparts, tpwgts, options, &objval, epart,
                                    npart);
    }
    if (status != METIS_OK) {
        fprintf(stderr, "METIS status %d\n", status);
        return 4;
    }
    const idx_t nout = nodal ? nn : ne;
    int32_t *out = malloc(sizeof(int32_t) * nout);
    for (idx_t e = 0; e < nout; ++e) out[e] = (int32_t)(nodal ? npart[e] : epart[e]);
    f = fopen(argv[4], "wb");
    fwrite(out, sizeof(int32_t), nout, f);
    fclose(f);
    fprintf(stderr, "metis_part: ne=%ld nn=%ld nparts=%ld edgecut=%ld\n", (long)ne, (long)nn,
            (long)nparts, (long)objval);
    return 0;
}
