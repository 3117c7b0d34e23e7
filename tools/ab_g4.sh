for cfg in "--method 9 --rays 524288 --record none --steps 3" "--method 5 --rays 524288 --record none --steps 3" "--scenario anisotropy --record none --steps 3" "--scenario fisheye --method 9 --rays 524288 --record none --steps 3"; do
  for lib in raytracing_amd/librtmi.so build/librtmi_g4.so; do
    echo -n "$(basename $lib) : "
    RTMI_LIB_PATH=$PWD/$lib python3 tools/bench_line.py $cfg
  done
done
