"""MI355X-native JPEG decode/encode extension behind the nvImageCodec plugin C-ABI."""
