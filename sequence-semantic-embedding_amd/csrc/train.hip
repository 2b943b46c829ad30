// Training kernels: placeholder until the backward path lands.
#include "train.h"
struct TrainState { int unused; };
void train_state_free(TrainState *t) { delete t; }
