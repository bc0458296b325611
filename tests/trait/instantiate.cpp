// TEST INFRASTRUCTURE (compile-only, authoring container): the registration unit a maintainer would add as
// src/impl/VeryFastTree{Float,Double}Hip.cpp - the reference's whole pipeline template
// instantiated with HipOperations in the Operations slot (NeighbourJoining.h:19-22, VeyFastTreeImpl.h:17-26).
// Includes the reference headers where they lie; contains no reference source.
#include "Utils.h"
#include "HipOperations.h"
#include "VeyFastTreeImpl.h"

template class veryfasttree::VeyFastTreeImpl<float, veryfasttree::HipOperations>;
template class veryfasttree::VeyFastTreeImpl<double, veryfasttree::HipOperations>;
