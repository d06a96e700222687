/* TEST INFRASTRUCTURE ONLY — plain-C restatement of SimpleFC.forward.
 *
 * Follows /root/reference/utils/nn_model.py:21-33 (layer list: Linear, LeakyReLU(0.01), Dropout
 * for every hidden layer; then Linear, Sigmoid) and :38-41 (sequential apply). Dropout is the
 * identity at eval (the reference calls model.eval(), _5_predict_labels.py:108).
 * Pinned against the reference itself: tests/golden/make_golden.py imports the reference
 * SimpleFC + the shipped checkpoint and stores its outputs in tests/golden/regressor_*.npz.
 *
 * Built by oracle/Makefile into oracle/_build/libfcreg_oracle.so; only tests/, smoke() and the
 * cpu_baseline leg of bench.py may load it.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* sizes[0..n_layers] : in, hidden..., out ; W[l] is row-major [sizes[l+1]][sizes[l]] (nn.Linear) */
int fcreg_oracle_forward(int n_layers, const int *sizes, const float *const *W,
                         const float *const *b, float negative_slope,
                         const float *x, int n_rows, float *y)
{
    int max_w = 0;
    for (int l = 0; l <= n_layers; ++l) if (sizes[l] > max_w) max_w = sizes[l];
    float *cur = (float *)malloc(sizeof(float) * (size_t)max_w);
    float *nxt = (float *)malloc(sizeof(float) * (size_t)max_w);
    if (!cur || !nxt) { free(cur); free(nxt); return 1; }
    const int out = sizes[n_layers];
    for (int r = 0; r < n_rows; ++r) {
        memcpy(cur, x + (size_t)r * sizes[0], sizeof(float) * (size_t)sizes[0]);
        for (int l = 0; l < n_layers; ++l) {
            const int in = sizes[l], on = sizes[l + 1];
            for (int j = 0; j < on; ++j) {
                const float *w = W[l] + (size_t)j * in;
                float acc = 0.0f;
                for (int k = 0; k < in; ++k) acc += w[k] * cur[k];
                acc += b[l][j];
                if (l < n_layers - 1) acc = acc >= 0.0f ? acc : negative_slope * acc;   /* LeakyReLU */
                else acc = 1.0f / (1.0f + expf(-acc));                                  /* Sigmoid */
                nxt[j] = acc;
            }
            float *t = cur; cur = nxt; nxt = t;
        }
        memcpy(y + (size_t)r * out, cur, sizeof(float) * (size_t)out);
    }
    free(cur); free(nxt);
    return 0;
}
