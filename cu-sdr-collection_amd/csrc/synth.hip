// synth.hip — placeholder (GPU IF synthesiser lands with bench.py).
extern "C" int gs_version(void) { return 0; }
