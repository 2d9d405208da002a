// cli.hpp -- the reference's command line (src/bin/lightdock-rust.rs:77-333) on the HIP engine.
#pragma once
namespace ld {
int cli_main(int argc, char **argv);
}
