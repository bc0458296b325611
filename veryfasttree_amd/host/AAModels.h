// Amino-acid models as the reference sets them up, on the host: the three built-in exchange matrices
// (createTransitionMatrixJTT92 / WAG01 / LG08, TransitionMatrix.tcc:14-24 -> createTransitionMatrix :158-232), the
// default BLOSUM45-derived distance matrix of the NJ / minimum-evolution phase (matrixBLOSUM45 + setupDistanceMatrix,
// DistanceMatrix.tcc:33-36, 102-155) and the transition matrix re-expressed as a "distance matrix" for re-averaging
// the profiles before the ML phase (transMatToDistanceMat, VeryFastTreeImpl.tcc:517-542).  The constants live in
// AAModelData.h (generated); the tables cross the boundary through vft_set_transition_matrix / vft_set_distance_matrix.
#ifndef VFT_AA_MODELS_H
#define VFT_AA_MODELS_H

#include <stdexcept>

#include "AAModelData.h"
#include "GtrModel.h"

namespace veryfasttree {

    enum AAModel {
        AA_MODEL_JTT = 1,   /* the reference's default for amino acids (VeryFastTreeImpl.tcc:104-106) */
        AA_MODEL_WAG = 2,   /* -wag */
        AA_MODEL_LG = 3     /* -lg */
    };

    typedef TransitionTables<20> TransitionTables20;

    /* what vft_set_distance_matrix takes: numeric_t values held in double */
    struct DistanceTables20 {
        double distances[20][20], codeFreq[20][20], eigenval[20], eigentot[20];
    };

    template<typename REAL>
    inline void createAAModel(int model, TransitionTables20 &t) {
        const double *stat, *flat;
        switch (model) {
            case AA_MODEL_JTT:
                stat = aa_data::kStatJTT92;
                flat = aa_data::kMatrixJTT92;
                break;
            case AA_MODEL_WAG:
                stat = aa_data::kStatWAG01;
                flat = aa_data::kMatrixWAG01;
                break;
            case AA_MODEL_LG:
                stat = aa_data::kStatLG08;
                flat = aa_data::kMatrixLG08;
                break;
            default:
                throw std::invalid_argument("unknown amino-acid model (1 = JTT, 2 = WAG, 3 = LG)");
        }
        double matrix[20][20];
        for (int i = 0; i < 20; i++)
            for (int j = 0; j < 20; j++) matrix[i][j] = flat[20 * i + j];
        createTransitionTables<REAL, 20>(matrix, stat, t);
    }

    /* matrixBLOSUM45 + setupDistanceMatrix: the literals are narrowed to numeric_t where the reference's static
       initialiser narrows them; eigentot is summed in numeric_t (DistanceMatrix.tcc:127-133), codeFreq is the
       transpose of eigeninv (:135-140) */
    template<typename REAL>
    inline void blosum45Tables(DistanceTables20 &d) {
        REAL eigeninv[20][20];
        for (int i = 0; i < 20; i++) {
            d.eigenval[i] = (double) (REAL) aa_data::kBlosum45EigenVal[i];
            for (int j = 0; j < 20; j++) {
                d.distances[i][j] = (double) (REAL) aa_data::kBlosum45Distances[20 * i + j];
                eigeninv[i][j] = (REAL) aa_data::kBlosum45EigenInv[20 * i + j];
            }
        }
        for (int k = 0; k < 20; k++) {
            REAL tot = 0;
            for (int j = 0; j < 20; j++) tot += eigeninv[k][j];
            d.eigentot[k] = (double) tot;
        }
        for (int code = 0; code < 20; code++)
            for (int k = 0; k < 20; k++) d.codeFreq[code][k] = (double) eigeninv[k][code];
    }

    /* transMatToDistanceMat: rotation and normalisation of the transition matrix in the DistanceMatrix slots; the
       distances ("never actually used") and the eigenvalues stay zero as in the reference's value-initialised object */
    template<typename REAL>
    inline void transitionAsDistanceTables(const TransitionTables20 &t, DistanceTables20 &d) {
        for (int i = 0; i < 20; i++) {
            d.eigenval[i] = 0;
            REAL tot = 0;
            for (int j = 0; j < 20; j++) {
                d.distances[i][j] = 0;
                d.codeFreq[i][j] = t.codeFreq[i][j];
                tot += (REAL) t.eigeninv[i][j];
            }
            d.eigentot[i] = (double) tot;
        }
    }

}

#endif
