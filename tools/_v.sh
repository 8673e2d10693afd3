STREAMS=2 bash tools/bench_variants.sh t_b512 t_b768 t_b512s256 t_b512w16 t_b384
for gd in 2; do echo "grid_div $gd:"; PTX_GRID_DIV=$gd STREAMS=2 bash tools/bench_variants.sh t_b512 t_base; done
for ns in 3 4; do echo "streams $ns:"; STREAMS=$ns bash tools/bench_variants.sh t_b512; done
