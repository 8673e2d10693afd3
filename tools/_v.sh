for wgs in 0 4 3 2; do for v in c_base c_s256; do echo -n "trace_wgs $wgs: "; PTX_TRACE_WGS=$wgs STREAMS=2 bash tools/bench_variants.sh $v; done; done
